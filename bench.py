#!/usr/bin/env python3
"""bench.py -- headline benchmark of the NAF hot path on MI355X.

N = 1: the size BASELINE.json's metric is quoted on -- 100 GB of synthetic-ACGT FASTA -- on the one GPU it fits (text, archive and
    decoded text resident in HBM: 225 of the 288 GB).  One "step" = one complete naf_gpu_unnaf pass over the archive (small
    sections + offset scans + zstd decode of the sequence stream + 4-bit unpack / mask / line-wrap emit).  The archive is made
    in-run by the GPU encoder (naf_gpu_ennaf, timed on its own: `ennaf_value`); the decoded text is checked at full size (bit-exact
    in chunks, position-weighted checksum).  Behind the headline, with its buffers freed: BASELINE configs[1] / [2] at their own
    10 GB (`cfg10`, `ennaf_roofline` with the PMC traffic of that size), the decode of the REFERENCE-made archive of that text, a
    soft-masked text, a realistic genome, FASTQ at one GPU's share of configs[4] (12.5 GB), levels -19 / --long beside the reference
    with the same flags, the reference on this box's host cores and the file -> file times of both CLIs next to the reference's.
    The driver's record keeps the scalars of `roofline` and `cpu_baseline`: every side figure has a flat copy there.

N > 1 (configs[3]: the same 100 GB as ONE archive on N GPUs, size / N of the text per GPU -- strong scaling):
    the archive -- made in-run by the sharded encoder (naf_gpu_ennaf_shard_*: every rank encodes its slice, the parts are joined
    into one frame per stream; configs[4] shape) -- is decoded by all ranks: rank r produces bytes [r, r+1) x total/N of the text
    with naf_gpu_unnaf_range and the ranges are gathered to rank 0 over RCCL (one group of point-to-point transfers into place,
    posted before the root decodes its own range).  One step = range decode on every rank + the gather.  A gather-to-root is bound
    by the root's xGMI ingress, BELOW what one GPU decodes on its own (DESIGN.md section 6 has the predicted curve): the line carries
    `decode_only_value` and `to_host_value` (per-GPU D2H, PCIe-inclusive, never `value`) beside it.  `--replicas` runs the round-1
    shape instead (one archive per GPU, no data-path collective).

Emits ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, /opt/skills/guides/MI355X_MICROARCH.md
REF_E = os.path.join(ROOT, "oracle", "_ref", "ennaf")
REF_U = os.path.join(ROOT, "oracle", "_ref", "unnaf")
BIN = os.path.join(ROOT, "naf_amd", "bin")


def have_ref():
    return os.access(REF_E, os.X_OK) and os.access(REF_U, os.X_OK)


def box_id():
    """A short stamp of the GPU box this line was measured on (kernel times differ by up to 10 % from box to box): the device's unique id
    where the driver exposes one, else a hash of the host name.  tools/profile_bench.sh writes the same stamp beside the rocprofv3
    summaries it leaves under profiles/, so that a `frac` in the line and one recomputed from profiles/ can be told to be the same box's."""
    import glob
    import hashlib
    ids = []
    for f in sorted(glob.glob("/sys/class/drm/card*/device/unique_id")):
        try:
            ids.append(open(f).read().strip())
        except OSError:
            pass
    src = ids[0] if ids and ids[0] else os.uname().nodename
    return hashlib.sha256(src.encode()).hexdigest()[:10]


def timed_calls(fn, reps, warm=2):
    """fn() `warm` times untimed (a context's arenas grow in the first call that needs them and are one allocation when that call
    returns -- a second untimed call is there so that nothing of it is timed whatever the library did), then `reps` calls each timed on
    its own between two device synchronisations.  Returns the list of seconds."""
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return ts


def stats_ms(ts):
    """Median, min, max and mean of per-call times: the median is the figure (one host hiccup in a window of three used to double a mean)."""
    v = sorted(ts)
    n = len(v)
    med = v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])
    return {"median_ms": round(med * 1e3, 3), "min_ms": round(v[0] * 1e3, 3), "max_ms": round(v[-1] * 1e3, 3), "mean_ms": round(sum(v) / n * 1e3, 3), "reps": n}


def median(ts):
    v = sorted(ts)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def instrumented(ctx, fn, top=6):
    """One more call with the library's per-kernel HIP events on: the longest kernels and the kernel time per stream of the context."""
    ctx.set_timing(True)
    fn()
    kt = ctx.get_timing()
    streams = ctx.get_timing_streams()
    ctx.set_timing(False)
    return sorted(kt, key=lambda x: -x[1])[:top], kt, streams


def gap_of(call_ms, streams):
    """What the call takes beyond its busiest stream's kernels: launch gaps, host read-backs, waits between streams."""
    return {"gap_ms": round(call_ms - max(streams), 3), "stream_kernel_ms": [round(x, 3) for x in streams]}


def last_line_end(t):
    """Length of the longest prefix of t that ends with a line end (looked for in its last 64 KiB)."""
    w = min(int(t.numel()), 65536)
    tail = t[t.numel() - w:]
    return int(t.numel()) - w + int((tail == 10).nonzero()[-1].item()) + 1


def cpu_baseline(text_dev, size_bytes, ctx=None, full_ref_archive=True, e2e_bytes=0, reps=10):
    """Reference unnaf/ennaf (oracle/_ref, built from the reference sources) on this box's host cores, one thread (the reference
    is single-threaded), on a bounded sample of the same workload; the GPU decoder on the archive the reference makes of the
    WHOLE text; and file -> file wall time of both CLIs beside the reference's."""
    import numpy as np
    if not have_ref():
        return None
    shm = "/dev/shm/naf_bench_%d" % os.getpid()
    os.makedirs(shm, exist_ok=True)
    P = lambda name: os.path.join(shm, name)
    try:
        sample = text_dev[:size_bytes]
        # cut at a line end so the sample is a well-formed FASTA prefix
        cut = last_line_end(sample)
        sample[:cut].cpu().numpy().tofile(P("s.fa"))
        env = dict(os.environ, TMPDIR=shm)
        t0 = time.perf_counter()
        subprocess.check_call([REF_E, P("s.fa"), "-o", P("s.naf")], env=env)
        t_e = time.perf_counter() - t0
        t0 = time.perf_counter()
        subprocess.check_call([REF_U, P("s.naf"), "-o", P("s.out")])
        t_u = time.perf_counter() - t0
        same = subprocess.call(["cmp", "-s", P("s.fa"), P("s.out")]) == 0
        out = {"value": round(cut / t_u / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "reference",
               "sample": "reference unnaf (oracle/_ref, libzstd 1.4.9) on the first %.2f GB of the same FASTA, tmpfs, 1 thread" % (cut / 1e9),
               "ennaf_value": round(cut / t_e / 1e9, 4), "roundtrip_ok": bool(same)}
        # the "whole box" figure -- the reference is single-threaded, so one instance per host core (capped), all
        # decoding the same archive concurrently to /dev/null; aggregate = instances x sample bytes / wall time of the slowest
        ncpu = min(os.cpu_count() or 1, 64)
        if ncpu > 1:
            t0 = time.perf_counter()
            with open(os.devnull, "wb") as dn:
                ps = [subprocess.Popen([REF_U, P("s.naf")], stdout=dn) for _ in range(ncpu)]
                rcs = [q.wait() for q in ps]
            t_all = time.perf_counter() - t0
            if all(rc == 0 for rc in rcs):
                out["all_cores"] = {"value": round(ncpu * cut / t_all / 1e9, 3), "unit": "GB/s", "cores": ncpu,
                                    "sample": "%d concurrent reference unnaf instances, each decoding the same %.2f GB sample archive to /dev/null" % (ncpu, cut / 1e9)}
        if ctx is not None:
            # the GPU decoder on the archive the REFERENCE ennaf makes (128 KiB dependent blocks) -- of the whole text
            import torch
            from naf_amd import capi
            src, n_src, naf_path = P("s.fa"), cut, P("s.naf")
            if full_ref_archive and text_dev.numel() > cut:
                try:
                    text_dev.cpu().numpy().tofile(P("full.fa"))
                    t0 = time.perf_counter()
                    subprocess.check_call([REF_E, P("full.fa"), "-o", P("full.naf")], env=env)
                    n_src, naf_path = int(text_dev.numel()), P("full.naf")
                    out["ennaf_full"] = {"value": round(n_src / (time.perf_counter() - t0) / 1e9, 4), "unit": "GB/s", "text_bytes": n_src}
                except (OSError, subprocess.CalledProcessError) as ex:             # tmpfs too small for the whole text: the sample's archive
                    out["full_archive_skipped"] = repr(ex)[:120]
                if os.path.exists(P("full.fa")):
                    os.remove(P("full.fa"))
            ref_naf = torch.from_numpy(np.fromfile(naf_path, dtype=np.uint8)).to(text_dev.device)
            buf = torch.empty(n_src + 64, dtype=torch.uint8, device=text_dev.device)
            r = ctx.unnaf(ref_naf, capi.OUT_FASTA, out=buf)
            torch.cuda.synchronize()
            ok = bool(torch.equal(r, text_dev[:n_src]))
            ts = timed_calls(lambda: ctx.unnaf(ref_naf, capi.OUT_FASTA, out=buf), reps)
            dt = median(ts)
            kt, _all, streams = instrumented(ctx, lambda: ctx.unnaf(ref_naf, capi.OUT_FASTA, out=buf))
            # one eighth of the text by byte range (what a rank of an 8-GPU job decodes, BASELINE configs[3]): the frame has matches, so
            # the range's dependency closure is decoded (zstd_dec.hip: k_range_closure), not the whole stream
            b8 = n_src * 3 // 8; e8 = b8 + n_src // 8
            rbuf = torch.empty(e8 - b8 + 64, dtype=torch.uint8, device=text_dev.device)
            r8 = ctx.unnaf_range(ref_naf, b8, e8, capi.OUT_FASTA, out=rbuf); torch.cuda.synchronize()
            ok8 = bool(torch.equal(r8, text_dev[b8:e8]))
            ts8 = timed_calls(lambda: ctx.unnaf_range(ref_naf, b8, e8, capi.OUT_FASTA, out=rbuf), reps)
            dt8 = median(ts8)
            res = {"value": round(n_src / dt / 1e9, 2), "unit": "GB/s (median of %d calls, each timed on its own)" % reps, "ms": round(dt * 1e3, 3)}
            res.update(stats_ms(ts)); res.update(gap_of(dt * 1e3, streams))
            res.update({"path_frac": round((n_src + int(ref_naf.numel())) / dt / HBM_PEAK, 4),
                        "archive_bytes": int(ref_naf.numel()), "text_bytes": int(n_src), "bit_exact": ok,
                        "kernels_ms": {n: round(ms, 3) for n, ms, k in kt},
                        "range_eighth_ms": round(dt8 * 1e3, 3), "range_eighth": stats_ms(ts8), "range_eighth_bit_exact": ok8})
            out["gpu_unnaf_of_reference_archive"] = res
            del rbuf, r8
            del ref_naf, buf
        if e2e_bytes and os.access(os.path.join(BIN, "ennaf"), os.X_OK):
            # drop-in reality check: file -> file through the CLIs (tmpfs; PCIe, file I/O and process start included) -- never the `value`
            e2e = text_dev[:e2e_bytes]
            cut2 = last_line_end(e2e)
            e2e[:cut2].cpu().numpy().tofile(P("e.fa"))
            del e2e
            # the CLIs are processes of their own on this GPU: this process gives back what it holds beside the text (its context's
            # arenas, torch's cached blocks) before they start
            if ctx is not None:
                import torch
                torch.cuda.synchronize(); ctx.release_scratch(); torch.cuda.empty_cache()
            def timed(cmd, extra_env=None):
                t0 = time.perf_counter()
                rc = subprocess.call(cmd, env=dict(env, **(extra_env or {})), stderr=subprocess.DEVNULL)
                return round(time.perf_counter() - t0, 3) if rc == 0 else None
            res = {"text_bytes": cut2, "unit": "s, file -> file on tmpfs, to the exit of the process the caller started"}
            # each CLI twice, into a new file both times; the figure is the faster run, both are listed (the first process on a box that has
            # just finished other work pays for a cold driver: 0.2 - 0.4 s more in "GPU init").  Default = ONE process, timed to its exit
            # like the reference; NAF_GPU_DETACH=1 = the foreground process leaves when the output is complete and a worker's teardown goes
            # on behind it (host_common.h: detach_teardown) -- that worker is waited for before the next command is timed.
            def twice(cmd, out, extra_env=None):
                ts = []
                for _ in range(2):
                    if os.path.exists(out):
                        os.remove(out)
                    ts.append(timed(cmd, extra_env))
                    if extra_env:
                        time.sleep(0.5)
                return ts
            best = lambda ts: min(t for t in ts if t is not None) if any(t is not None for t in ts) else None
            ts = twice([os.path.join(BIN, "ennaf"), P("e.fa"), "-o", P("e.naf")], P("e.naf"))
            res["ennaf"] = best(ts); res["ennaf_runs"] = ts
            ts = twice([os.path.join(BIN, "unnaf"), P("e.naf"), "-o", P("e.out")], P("e.out"))
            res["unnaf"] = best(ts); res["unnaf_runs"] = ts
            t0 = time.perf_counter()
            with open(os.devnull, "wb") as dn:
                rc = subprocess.call([os.path.join(BIN, "unnaf"), P("e.naf"), "-c"], stdout=dn, stderr=subprocess.DEVNULL, env=env)
            res["unnaf_to_devnull"] = round(time.perf_counter() - t0, 3) if rc == 0 else None
            det = {"NAF_GPU_DETACH": "1"}
            ts = twice([os.path.join(BIN, "ennaf"), P("e.fa"), "-o", P("e.naf3")], P("e.naf3"), det)
            res["ennaf_detached"] = best(ts); res["ennaf_detached_runs"] = ts
            res["detached_archive_same"] = subprocess.call(["cmp", "-s", P("e.naf"), P("e.naf3")]) == 0
            if os.path.exists(P("e.naf3")):
                os.remove(P("e.naf3"))
            ts = twice([os.path.join(BIN, "unnaf"), P("e.naf"), "-o", P("e.out")], P("e.out"), det)
            res["unnaf_detached"] = best(ts); res["unnaf_detached_runs"] = ts
            res["roundtrip_ok"] = subprocess.call(["cmp", "-s", P("e.fa"), P("e.out")]) == 0
            same = lambda a, b: subprocess.call(["cmp", "-s", P(a), P(b)]) == 0
            # the drop-in direction at this size: the REFERENCE's streaming loop (unnaf/src/output.c:640-651) on the archive this build made
            res["reference_unnaf_of_gpu_archive"] = timed([REF_U, P("e.naf"), "-o", P("e.out2")])
            res["reference_unnaf_of_gpu_archive_bit_exact"] = same("e.fa", "e.out2")
            os.remove(P("e.out2"))
            res["reference_ennaf"] = timed([REF_E, P("e.fa"), "-o", P("e.ref.naf")])
            res["reference_unnaf"] = timed([REF_U, P("e.ref.naf"), "-o", P("e.out")])
            res["reference_roundtrip_ok"] = same("e.fa", "e.out")
            os.remove(P("e.out"))
            res["unnaf_of_reference_archive"] = timed([os.path.join(BIN, "unnaf"), P("e.ref.naf"), "-o", P("e.out")])
            res["unnaf_of_reference_archive_bit_exact"] = same("e.fa", "e.out")
            # where a CLI's wall time goes (the hosts print their phases under NAF_GPU_PHASES=1)
            for f in ("e.out", "e.naf2"):
                if os.path.exists(P(f)):
                    os.remove(P(f))
            res["phases"] = {"ennaf": cli_phases([os.path.join(BIN, "ennaf"), P("e.fa"), "-o", P("e.naf2")], env),
                             "unnaf": cli_phases([os.path.join(BIN, "unnaf"), P("e.naf"), "-o", P("e.out")], env),
                             "unnaf_to_devnull": cli_phases([os.path.join(BIN, "unnaf"), P("e.naf"), "-o", "/dev/null"], env)}
            out["end_to_end"] = res
            # flat copies for the driver's record (it keeps the scalars of this object)
            out.update({"e2e_text_bytes": cut2, "e2e_unnaf_s": res["unnaf"], "e2e_ennaf_s": res["ennaf"], "e2e_unnaf_s_detached": res["unnaf_detached"],
                        "e2e_ennaf_s_detached": res["ennaf_detached"], "e2e_unnaf_to_devnull_s": res["unnaf_to_devnull"],
                        "e2e_ref_unnaf_s": res["reference_unnaf"], "e2e_ref_ennaf_s": res["reference_ennaf"], "e2e_roundtrip_ok": bool(res["roundtrip_ok"]),
                        "e2e_ref_decodes_gpu_archive": bool(res["reference_unnaf_of_gpu_archive_bit_exact"]),
                        "e2e_gpu_decodes_ref_archive": bool(res["unnaf_of_reference_archive_bit_exact"])})
        ac = out.get("all_cores") or {}
        out["all_cores_gbps"] = ac.get("value"); out["all_cores_n"] = ac.get("cores")
        return out
    finally:
        subprocess.call(["rm", "-rf", shm])


def cli_phases(cmd, env):
    """Wall time of the phases a CLI host reports under NAF_GPU_CLI_TIMING=1 (host_common.h: phase), as {phase: ms}."""
    r = subprocess.run(cmd, env=dict(env, NAF_GPU_CLI_TIMING="1"), stderr=subprocess.PIPE, stdout=subprocess.DEVNULL)
    out = {}
    for ln in r.stderr.decode("latin1").splitlines():
        if ln.startswith("[timing] ") and ln.rstrip().endswith(" ms"):
            name, ms = ln[9:].rstrip()[:-3].rsplit(None, 1)
            out[name.strip()] = out.get(name.strip(), 0.0) + float(ms)
    return {k: round(v, 1) for k, v in out.items()} if r.returncode == 0 else None


def fastq_same_but_case(back, text):
    """FASTQ comes back with upper-case bases (unnaf.c:442): the SEQUENCE lines may differ in the case bit of a letter, every
    other byte -- names, '+' lines, qualities, line ends -- must be identical.  Line ordinals from a running count of line ends."""
    import torch
    n = int(text.numel())
    if int(back.numel()) != n:
        return False
    step = 1 << 27
    lines = 0
    for a in range(0, n, step):
        x, y = back[a:a + step], text[a:a + step]
        eol = (y == 10)
        # ordinal of the line a byte lies in = line ends in front of it
        ordn = torch.cumsum(eol.to(torch.int32), 0, dtype=torch.int64) - eol.to(torch.int64) + lines
        seq_line = (ordn & 3) == 1
        letter = ((y | 32) >= 97) & ((y | 32) <= 122)
        ok = (x == y) | (seq_line & letter & ((x ^ 32) == y))
        if not bool(ok.all()):
            return False
        lines += int(eol.sum().item())
        del ordn, seq_line, letter, ok, eol
    return True


TRAFFIC_FILES = {"pmc_traffic.json": ("pmc_traffic_100gb.json", "pmc_traffic.json"), "pmc_traffic_ennaf.json": ("pmc_traffic_ennaf_100gb.json", "pmc_traffic_ennaf.json")}


def traffic_file(n_text, fname):
    """The committed PMC summary (tools/profile_bench.sh) whose run had this text size: HBM bytes per kernel and call."""
    for f in TRAFFIC_FILES.get(fname, (fname,)):
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", f)))
            if abs(pm["text_bytes"] - n_text) < 0.01 * n_text:
                return pm
        except (OSError, KeyError, ValueError):
            pass
    return None


def load_traffic(kernel, n_text, fname="pmc_traffic.json"):
    pm = traffic_file(n_text, fname)
    if pm and kernel in pm["kernels"]:
        return int(pm["kernels"][kernel]["fetch_bytes"] + pm["kernels"][kernel]["write_bytes"]), pm["source"]
    return None, None


def load_traffic_sum(n_text, fname):
    """HBM bytes of a whole call: every kernel of the PMC summary."""
    pm = traffic_file(n_text, fname)
    return int(sum(k["fetch_bytes"] + k["write_bytes"] for k in pm["kernels"].values())) if pm else None


def roofline_of(kt, alg, n_text, path_bytes, ms_per_step, fname="pmc_traffic.json", merge_side=()):
    """Roofline object of the dominant kernel among `alg` (name -> algorithmic bytes per step).  Launches on the side contexts'
    streams are reported as "side:<name>"; a kernel in `merge_side` counts its side launches too (the emit of a split decode)."""
    def kernel_ms(name):
        ms, k = kt.get(name, (0.0, 0))
        if name in merge_side:
            ms2, k2 = kt.get("side:" + name, (0.0, 0)); ms += ms2; k += k2
        return ms, max(k, 1)
    dom = max(alg, key=lambda k: kernel_ms(k)[0])
    dom_ms, dom_launches = kernel_ms(dom)
    if dom_ms <= 0:
        return None
    achieved = alg[dom] / (dom_ms * 1e-3) / 1e9
    traffic, traffic_src = load_traffic(dom, n_text, fname)
    return {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
            "frac": round(achieved * 1e9 / HBM_PEAK, 4), "traffic": traffic, "traffic_source": traffic_src,
            # the counters cannot be read from inside this process: `traffic` is the figure of the committed rocprofv3 --pmc passes of this
            # same command at this text size (tools/profile_bench.sh), not of this run
            "traffic_replayed": traffic is not None,
            "algorithmic_bytes_per_launch": int(alg[dom] // dom_launches), "avg_launch_ms": round(dom_ms / dom_launches, 4),
            "kernel_ms_per_step": round(dom_ms, 4), "launches_per_step": dom_launches,
            "path_bytes_per_step": int(path_bytes),
            "path_frac": round(path_bytes / (ms_per_step * 1e-3) / HBM_PEAK, 4),
            "kernels_ms": {n: round(ms, 3) for n, (ms, k) in sorted(kt.items(), key=lambda x: -x[1][0])[:8]}}


def weighted_sum(t, offset):
    """Position-weighted byte sum (fits int64 up to 100 GB): additive over consecutive ranges, so the sum over the ranks' ranges
    of the decoded text must equal the sum over the ranks' slices of the input -- a checksum of checksums at full size."""
    import torch
    if t.numel() == 0:
        return 0
    tot = 0
    step = 1 << 28
    for a in range(0, t.numel(), step):
        seg = t[a:a + step].to(torch.int64)
        idx = (torch.arange(a, a + seg.numel(), device=t.device, dtype=torch.int64) + offset) % 65521 + 1
        tot += int((seg * idx).sum().item())
    return tot


def realistic_vs_reference(ctx, text_dev, size_bytes, out_mode=None, fold_case=False):
    """The realistic genome against the real reference on a bounded sample: the reference's archive of the sample decoded by the GPU
    (bit-exact, timed, per kernel) and by the reference itself (one thread), and the GPU's archive of the sample decoded by the
    reference (the drop-in direction)."""
    import numpy as np
    import torch
    from naf_amd import capi
    shm = "/dev/shm/naf_bench_rg_%d" % os.getpid()
    os.makedirs(shm, exist_ok=True)
    P = lambda name: os.path.join(shm, name)
    try:
        cut = last_line_end(text_dev[:size_bytes])
        if fold_case:
            # FASTQ: the sample ends with a whole record -- in front of the last line that starts a record ("\n@read")
            w = min(cut, 8192)
            tail = text_dev[cut - w:cut].cpu().numpy().tobytes()
            k = tail.rfind(b"\n@read")
            cut = cut - w + k + 1
        sample = text_dev[:cut]
        sample.cpu().numpy().tofile(P("r.fa"))
        env = dict(os.environ, TMPDIR=shm)
        t0 = time.perf_counter(); subprocess.check_call([REF_E, P("r.fa"), "-o", P("r.naf")], env=env); t_e = time.perf_counter() - t0
        t0 = time.perf_counter(); subprocess.check_call([REF_U, P("r.naf"), "-o", P("r.out")]); t_u = time.perf_counter() - t0
        ref_naf = torch.from_numpy(np.fromfile(P("r.naf"), dtype=np.uint8)).to(text_dev.device)
        buf = torch.empty(cut + 64, dtype=torch.uint8, device=text_dev.device)
        mode = capi.OUT_FASTA if out_mode is None else out_mode
        r = ctx.unnaf(ref_naf, mode, out=buf); torch.cuda.synchronize()
        # the reference's own output of its archive is what "bit exact" means (FASTQ comes back with upper-case bases: unnaf.c:442)
        want = torch.from_numpy(np.fromfile(P("r.out"), dtype=np.uint8)).to(text_dev.device)
        ok = bool(torch.equal(r, want))
        del want
        ts = timed_calls(lambda: ctx.unnaf(ref_naf, mode, out=buf), 10)
        dt = median(ts)
        kt, _all, streams = instrumented(ctx, lambda: ctx.unnaf(ref_naf, mode, out=buf))
        mine, _rep = ctx.ennaf(sample)
        mine.cpu().numpy().tofile(P("m.naf"))
        subprocess.check_call([REF_U, P("m.naf"), "-o", P("m.out")])
        drop_in = subprocess.call(["cmp", "-s", P("r.out") if fold_case else P("r.fa"), P("m.out")]) == 0
        return {"reference_sample": {"text_bytes": int(cut), "reference_archive_bytes": int(ref_naf.numel()), "gpu_archive_bytes": int(mine.numel()),
                                     "reference_unnaf_value": round(cut / t_u / 1e9, 3), "reference_ennaf_value": round(cut / t_e / 1e9, 3),
                                     "gpu_unnaf_of_reference_archive": {"value": round(cut / dt / 1e9, 2), "ms": round(dt * 1e3, 3), "calls": stats_ms(ts), "bit_exact": ok,
                                                                        "gap": gap_of(dt * 1e3, streams), "kernels_ms": {n: round(ms, 3) for n, ms, k in kt}},
                                     "reference_unnaf_of_gpu_archive_bit_exact": bool(drop_in), "unit": "GB/s of text"}}
    finally:
        subprocess.call(["rm", "-rf", shm])


def reference_archive_at_size(ctx, text_dev, out_mode, fold_case=False, reps=5):
    """The REFERENCE's archive of the whole text of a side workload (the real `ennaf`, one thread: about half a minute for 12.5 GB of reads)
    decoded here: what a drop-in `unnaf` meets, at the size the config names rather than at a sample where a call is its latency
    (unnaf/src/input.c:145-246,295-434).  Bit-exact against the text (FASTQ: up to the case of the bases, unnaf.c:442)."""
    import numpy as np
    import torch
    shm = "/dev/shm/naf_bench_big_%d" % os.getpid()
    os.makedirs(shm, exist_ok=True)
    P = lambda name: os.path.join(shm, name)
    try:
        n = int(text_dev.numel())
        with open(P("t.txt"), "wb") as f:
            for a in range(0, n, 1 << 30):
                text_dev[a:a + (1 << 30)].cpu().numpy().tofile(f)
        env = dict(os.environ, TMPDIR=shm)
        t0 = time.perf_counter(); subprocess.check_call([REF_E, P("t.txt"), "-o", P("t.naf")] if not fold_case else [REF_E, "--fastq", P("t.txt"), "-o", P("t.naf")], env=env); t_e = time.perf_counter() - t0
        os.remove(P("t.txt"))
        ref_naf = torch.from_numpy(np.fromfile(P("t.naf"), dtype=np.uint8)).to(text_dev.device)
        os.remove(P("t.naf"))
        buf = torch.empty(n + 64, dtype=torch.uint8, device=text_dev.device)
        r = ctx.unnaf(ref_naf, out_mode, out=buf); torch.cuda.synchronize()
        ok = fastq_same_but_case(r, text_dev) if fold_case else equal_in_chunks(r, text_dev)
        ts = timed_calls(lambda: ctx.unnaf(ref_naf, out_mode, out=buf), reps)
        dt = median(ts)
        kt, _all, streams = instrumented(ctx, lambda: ctx.unnaf(ref_naf, out_mode, out=buf))
        return {"text_bytes": n, "reference_archive_bytes": int(ref_naf.numel()), "reference_ennaf_value": round(n / t_e / 1e9, 3),
                "value": round(n / dt / 1e9, 2), "ms": round(dt * 1e3, 3), "calls": stats_ms(ts), "bit_exact": bool(ok),
                "path_frac": round((n + int(ref_naf.numel())) / dt / HBM_PEAK, 4), "gap": gap_of(dt * 1e3, streams),
                "kernels_ms": {k: round(ms, 3) for k, ms, c in kt}, "unit": "GB/s of text out, the reference's own archive of the whole text in (device-resident)"}
    except (OSError, subprocess.CalledProcessError, capi_error()) as ex:
        return {"error": repr(ex)[:200]}
    finally:
        subprocess.call(["rm", "-rf", shm])


def capi_error():
    from naf_amd import capi
    return capi.NafGpuError


def side_workload(ctx, text, out_mode, what, fold_case=False, reps=10):
    """One more workload beside the headline config, device-resident both ways: ennaf then unnaf of `text`, each as `reps` calls timed on
    their own after two untimed ones (median = the figure; min, max and mean beside it), the round trip checked at full size, per-kernel
    device time of one instrumented call each way with what the call takes beyond its busiest stream, and the whole call against the
    HBM roofline on its algorithmic bytes (text + .naf)."""
    import torch
    from naf_amd import capi
    n = int(text.numel())
    nbuf = torch.empty(int(ctx.L.naf_gpu_ennaf_bound(n)), dtype=torch.uint8, device=text.device)
    res_e = [None]
    def enc():
        res_e[0] = ctx.ennaf(text, out=nbuf)
    ts_e = timed_calls(enc, reps)
    t_e = median(ts_e)
    ekt, _all, estreams = instrumented(ctx, enc, top=5)
    naf = res_e[0][0].clone(); del nbuf
    out = torch.empty(n + 64, dtype=torch.uint8, device=text.device)
    back = ctx.unnaf(naf, out_mode, out=out)
    torch.cuda.synchronize()
    ok = fastq_same_but_case(back, text) if fold_case else bool(torch.equal(back, text))
    ts_d = timed_calls(lambda: ctx.unnaf(naf, out_mode, out=out), reps)
    t_d = median(ts_d)
    dkt, _all, dstreams = instrumented(ctx, lambda: ctx.unnaf(naf, out_mode, out=out))
    n_naf = int(naf.numel())
    res = {"what": what, "text_bytes": n, "naf_bytes": n_naf, "naf_ratio": round(n_naf / n, 4), "unit": "GB/s of text (device-resident, median of %d calls each timed on its own)" % reps,
           "unnaf_value": round(n / t_d / 1e9, 3), "unnaf_ms": round(t_d * 1e3, 3), "unnaf_calls": stats_ms(ts_d), "unnaf_path_frac": round((n + n_naf) / t_d / HBM_PEAK, 4),
           "unnaf_kernels_ms": {k: round(ms, 3) for k, ms, c in dkt}, "unnaf_gap": gap_of(t_d * 1e3, dstreams),
           "ennaf_value": round(n / t_e / 1e9, 3), "ennaf_ms": round(t_e * 1e3, 3), "ennaf_calls": stats_ms(ts_e), "ennaf_path_frac": round((n + n_naf) / t_e / HBM_PEAK, 4),
           "ennaf_kernels_ms": {k: round(ms, 3) for k, ms, c in ekt}, "ennaf_gap": gap_of(t_e * 1e3, estreams),
           ("roundtrip_ok_case_folded" if fold_case else "roundtrip_bit_exact"): ok}
    del out, back
    return res, naf


def levels_leg(ctx, size, dev, sample_bytes=100_000_000):
    """the higher levels and `--long` get a number.  A repeat-rich genome of `size` bytes: GPU `ennaf -19` and
    `ennaf -3 --long 27` of the whole text (GB/s, median of three calls behind one untimed), archive sizes against the reference's with
    the same flags (ennaf/src/compressor.c:7-21, ennaf.c:247-273,505) -- level 19 on a bounded sample (libzstd at 19 packs ~3 MB/s), the
    window of 2^27 on the whole text -- and the real `unnaf` decoding this build's archives back to the text."""
    import numpy as np
    import torch
    from naf_amd import capi, synth
    shm = "/dev/shm/naf_bench_lv_%d" % os.getpid()
    os.makedirs(shm, exist_ok=True)
    P = lambda name: os.path.join(shm, name)
    env = dict(os.environ, TMPDIR=shm)
    out = {"what": "repeat-rich synthetic genome: 96 repeat families of 300..6000 bases over 45 % of the text, copies 0.3..6 % diverged, 60-column lines", "unit": "GB/s of text; ratio = this build's archive / the reference's with the same flags"}
    try:
        text = synth.repeat_genome_device(size, device=dev)
        n = int(text.numel())
        out["text_bytes"] = n
        text.cpu().numpy().tofile(P("t.fa"))
        cut = last_line_end(text[:min(n, sample_bytes)])
        text[:cut].cpu().numpy().tofile(P("s.fa"))
        nbuf = torch.empty(int(ctx.L.naf_gpu_ennaf_bound(n)), dtype=torch.uint8, device=dev)
        def gpu_enc(t, **kw):
            res = [None]
            def f():
                res[0] = ctx.ennaf(t, out=nbuf, **kw)
            ts = timed_calls(f, 3, warm=1)
            return res[0][0], median(ts)
        def ref_decodes(naf, want):
            naf.cpu().numpy().tofile(P("m.naf"))
            rc = subprocess.call([REF_U, P("m.naf"), "-o", P("m.out")], stderr=subprocess.DEVNULL)
            return rc == 0 and subprocess.call(["cmp", "-s", P(want), P("m.out")]) == 0
        # level 1 for scale
        a1, t1 = gpu_enc(text)
        out["lvl1_ennaf_gbps"] = round(n / t1 / 1e9, 3); out["lvl1_naf_bytes"] = int(a1.numel())
        # -19: speed on the whole text, ratio on the sample
        a19, t19 = gpu_enc(text, level=19)
        out["lvl19_ennaf_gbps"] = round(n / t19 / 1e9, 3); out["lvl19_ennaf_ms"] = round(t19 * 1e3, 2); out["lvl19_naf_bytes"] = int(a19.numel())
        s19 = ctx.ennaf(text[:cut], out=nbuf, level=19)[0]
        out["lvl19_sample_naf_bytes"] = int(s19.numel())
        out["lvl19_ref_decodes"] = bool(ref_decodes(s19, "s.fa"))
        t0 = time.perf_counter(); subprocess.check_call([REF_E, "--level", "19", P("s.fa"), "-o", P("s19.naf")], env=env); tr = time.perf_counter() - t0
        out["lvl19_ref_sample_naf_bytes"] = os.path.getsize(P("s19.naf")); out["lvl19_ref_ennaf_gbps"] = round(cut / tr / 1e9, 4); out["lvl19_sample_text_bytes"] = int(cut)
        out["lvl19_ratio_vs_ref"] = round(out["lvl19_sample_naf_bytes"] / out["lvl19_ref_sample_naf_bytes"], 4)
        # -3 --long 27: both on the whole text
        al, tl = gpu_enc(text, level=3, long_log=27)
        out["long27_ennaf_gbps"] = round(n / tl / 1e9, 3); out["long27_ennaf_ms"] = round(tl * 1e3, 2); out["long27_naf_bytes"] = int(al.numel())
        out["long27_ref_decodes"] = bool(ref_decodes(al, "t.fa"))
        t0 = time.perf_counter(); subprocess.check_call([REF_E, "--level", "3", "--long", "27", P("t.fa"), "-o", P("tl.naf")], env=env); tr = time.perf_counter() - t0
        out["long27_ref_naf_bytes"] = os.path.getsize(P("tl.naf")); out["long27_ref_ennaf_gbps"] = round(n / tr / 1e9, 4)
        out["long27_ratio_vs_ref"] = round(out["long27_naf_bytes"] / out["long27_ref_naf_bytes"], 4)
        # the GPU decoder on the reference's --long archive
        ref_l = torch.from_numpy(np.fromfile(P("tl.naf"), dtype=np.uint8)).to(dev)
        buf = torch.empty(n + 64, dtype=torch.uint8, device=dev)
        r = ctx.unnaf(ref_l, 0, out=buf); torch.cuda.synchronize()
        out["long27_ref_archive_bit_exact"] = bool(torch.equal(r, text))
        ts = timed_calls(lambda: ctx.unnaf(ref_l, 0, out=buf), 3, warm=1)
        out["long27_ref_archive_unnaf_gbps"] = round(n / median(ts) / 1e9, 2)
        t0 = time.perf_counter(); subprocess.check_call([REF_U, P("tl.naf"), "-o", P("tl.out")]); tr = time.perf_counter() - t0
        out["long27_ref_archive_ref_unnaf_gbps"] = round(n / tr / 1e9, 3)
        return out
    except (OSError, subprocess.CalledProcessError, capi.NafGpuError) as ex:
        out["error"] = repr(ex)[:200]
        return out
    finally:
        subprocess.call(["rm", "-rf", shm])


def equal_in_chunks(a, b, step=1 << 30):
    """torch.equal over slices of 1 GiB (the comparison of 100 GB at once would make a 100 GB boolean)."""
    import torch
    if int(a.numel()) != int(b.numel()):
        return False
    for p in range(0, int(a.numel()), step):
        if not bool(torch.equal(a[p:p + step], b[p:p + step])):
            return False
    return True


def one_gpu_leg(ctx, dev, size, records, seed, steps, warmup, enc_reps=10):
    """ennaf then unnaf of `size` bytes of synthetic-ACGT FASTA on this GPU, everything resident in HBM: the archive is made by the GPU
    encoder (timed on its own: `enc_reps` calls behind two untimed ones, one more instrumented), decoded once untimed and checked at full
    size (bit-exact in chunks + the position-weighted checksum), then `warmup` - 1 more untimed steps and EXACTLY `steps` timed ones between
    two device synchronisations, then one instrumented step.  Returns the measurements; text and archive stay alive in the result."""
    import torch
    from naf_amd import capi, synth
    text = synth.fasta_acgt_device(size, n_records=records, width=80, seed=seed, device=dev)
    n_text = int(text.numel())
    big = n_text > 40e9
    if not big:
        ctx.reserve(int(n_text * 1.7) + (2 << 30))     # scratch arena sized up front: growth (hipMalloc) and the consolidation after it are not part of a step
    naf_buf = torch.empty(int(n_text * 0.27) + (1 << 20), dtype=torch.uint8, device=dev)
    res_e = [None]
    def enc():
        res_e[0] = ctx.ennaf(text, out=naf_buf)
    enc_times = timed_calls(enc, enc_reps)                        # two untimed calls first (arena growth, lazy initialisation)
    d_naf, rep = res_e[0]
    _top, enc_all, enc_streams = instrumented(ctx, enc)
    free_after_encode = torch.cuda.mem_get_info()[0]
    if big:
        torch.cuda.synchronize(); ctx.release_scratch()        # the encoder's arena (the codes of 100 GB of text among it) before the text is made a second time
    out = torch.empty(n_text + 64, dtype=torch.uint8, device=dev)
    step = lambda: ctx.unnaf(d_naf, capi.OUT_FASTA, out=out)
    r = step()                                                    # untimed: the round trip at full size (size-independent properties)
    torch.cuda.synchronize()
    ok = equal_in_chunks(r, text)
    wsum = weighted_sum(text, 0)
    wsum_ok = wsum == weighted_sum(r, 0)
    for _ in range(max(0, warmup - 1)):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ctx.set_timing(True)
    step()
    kt = {n: (ms, k) for n, ms, k in ctx.get_timing()}
    step_streams = ctx.get_timing_streams()
    ctx.set_timing(False)
    del out, r
    return {"text": text, "d_naf": d_naf, "naf_buf": naf_buf, "rep": rep, "n_text": n_text, "n_naf": int(d_naf.numel()), "ok": ok, "wsum": wsum, "wsum_ok": wsum_ok,
            "dt": dt, "ms_per_step": dt / steps * 1e3, "value": n_text * steps / dt / 1e9, "kt": kt, "step_streams": step_streams,
            "enc_times": enc_times, "enc_kt": {n: (ms, k) for n, ms, k in enc_all}, "enc_streams": enc_streams, "free_after_encode": free_after_encode}


def decode_roofline(leg, ms_per_step, my_text=None, frac_mine=1.0, n_naf=None):
    """`roofline` of a decode step from its instrumented call: algorithmic bytes per step of each candidate dominant kernel (DESIGN.md
    section 3); a rank of a sharded job moves its share."""
    rep = leg["rep"]; n_text = leg["n_text"]
    my_text = n_text if my_text is None else my_text
    n_naf = leg["n_naf"] if n_naf is None else n_naf
    packed = (rep.n_bases + 1) // 2
    alg = {"zstd_huf_literals": (rep.section_comp[4] + packed) * frac_mine, "zstd_flat_literals": (rep.section_comp[4] + packed) * frac_mine,
           "unnaf_emit": packed * frac_mine + my_text,
           # a flat frame is read in place: compressed sequence stream in, text out, no packed intermediate
           "unnaf_emit_flat": rep.section_comp[4] * frac_mine + my_text}
    roofline = roofline_of(leg["kt"], alg, n_text, (n_naf * frac_mine + my_text), ms_per_step, merge_side=("unnaf_emit",))
    if roofline:
        roofline.update(gap_of(ms_per_step, leg["step_streams"]))
        roofline["box"] = box_id()
    return roofline


def encode_roofline(leg):
    rep = leg["rep"]; n_text = leg["n_text"]; n_naf = leg["n_naf"]
    T = rep.n_bases
    packed = (T + 1) // 2
    comp = rep.section_comp[4]
    # what the scatter pass has to move: the text in; out, the sequence stream's codes -- for direct blocks (pure ACGT: all of this
    # config) the FINAL 4-bit Huffman codes of packed pairs, 2 bits per base, not the packed bytes
    # (this text has no lower case: the count pass learns it and the scatter pass writes no case bits -- enc.hip: alloc_bases)
    ealg = {"ennaf_scatter_regular": n_text + T // 4, "ennaf_scatter": n_text + packed + T // 8, "ennaf_count_pure": n_text, "ennaf_count": n_text, "ennaf_last": n_text // 16,
            "ennaf_split_once": n_text + T // 4, "zenc_plan": packed, "zenc_write": packed + comp}
    enc_ms = median(leg["enc_times"]) * 1e3
    er = roofline_of(leg["enc_kt"], ealg, n_text, n_text + n_naf, enc_ms, fname="pmc_traffic_ennaf.json")
    if er:
        er["note"] = "frac prices the dominant kernel on its own bytes; path_frac = (text + .naf) / whole call / 8 TB/s is the encode's headline fraction (SURVEY 8(d))"
        er["calls"] = stats_ms(leg["enc_times"])
        er.update(gap_of(enc_ms, leg["enc_streams"]))
        tr = load_traffic_sum(n_text, "pmc_traffic_ennaf.json")
        if tr:
            er["call_traffic"] = tr; er["call_traffic_ratio"] = round(tr / float(n_text + n_naf), 3); er["traffic_replayed"] = True
    return er


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=float, default=100e9, help="FASTA bytes of the whole job: the metric's own 100 GB (N = 1: on the one GPU; N > 1: size / N per GPU, one archive)")
    ap.add_argument("--cfg10-size", type=float, default=10e9, help="N = 1: bytes of the BASELINE configs[1] / [2] leg beside the headline (0: skip)")
    ap.add_argument("--records", type=int, default=100)
    ap.add_argument("--cpu-sample", type=float, default=2e9, help="bytes of FASTA timed on the CPU reference")
    ap.add_argument("--e2e-size", type=float, default=4e9, help="bytes of FASTA run file -> file through the CLIs (0: skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--replicas", action="store_true",
                    help="N > 1: one archive per GPU, no data-path collective (the round-1 shape) instead of one sharded archive + gather")
    ap.add_argument("--force-sharded", action="store_true", help="run the N > 1 code path with whatever world size there is (1 on the test box)")
    ap.add_argument("--softmask-size", type=float, default=4e9, help="N = 1: bytes of a soft-masked FASTA (runs of 20..600 bases) encoded and decoded beside the headline config (0: skip)")
    ap.add_argument("--realistic-size", type=float, default=4e9, help="N = 1: bytes of a synthetic repeat-masked genome (skewed composition, N runs, IUPAC, soft mask) encoded and decoded beside the headline config (0: skip)")
    ap.add_argument("--fastq1-size", type=float, default=12.5e9, help="N = 1: bytes of cfg5 FASTQ encoded and decoded beside the headline config: one GPU's share of configs[4] (0: skip)")
    ap.add_argument("--fastq-size", type=float, default=12.5e9, help="N > 1: FASTQ bytes per GPU for the sharded FASTQ encode (configs[4] = 100 GB over 8 GPUs; 0: skip)")
    ap.add_argument("--no-ref-full", dest="ref_full", action="store_false", help="N = 1: skip the reference's archives of the WHOLE FASTQ / realistic texts (half a minute of the reference's ennaf)")
    ap.add_argument("--levels-size", type=float, default=1e9, help="N = 1: bytes of a repeat-rich genome encoded at -19 and -3 --long 27 beside the reference with the same flags (0: skip)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    multi = world > 1 or args.force_sharded
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29555")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    dev = "cuda:%d" % local
    sharded = multi and not args.replicas

    from naf_amd import capi, synth, shard
    ctx = capi.Context(local)
    extra = {}
    cfg10 = None
    cb = None
    ennaf_roofline = None

    if not multi:
        # ---- N = 1: the metric's own size on the one GPU (text, archive and decoded text resident: 225 of the 288 GB)
        leg = one_gpu_leg(ctx, dev, int(args.size), args.records, 2024, args.steps, args.warmup)
        n_text, n_naf, total_text, rep = leg["n_text"], leg["n_naf"], leg["n_text"], leg["rep"]
        value, ms_per_step, ok = leg["value"], leg["ms_per_step"], leg["ok"] and leg["wsum_ok"]
        enc_times = leg["enc_times"]
        roofline = decode_roofline(leg, ms_per_step)
        hl_enc = encode_roofline(leg)
        if roofline:
            roofline.update({"roundtrip_bit_exact": bool(leg["ok"]), "weighted_sum_equal": bool(leg["wsum_ok"]), "text_bytes": n_text, "naf_bytes": n_naf,
                             "hbm_free_after_encode_gb": round(leg["free_after_encode"] / 1e9, 1),
                             "ennaf_gbps": round(n_text / median(enc_times) / 1e9, 3), "ennaf_ms": round(median(enc_times) * 1e3, 3),
                             "ennaf_path_frac": round((n_text + n_naf) / median(enc_times) / HBM_PEAK, 4)})
        extra["headline_ennaf_roofline"] = hl_enc
        extra["weighted_sum"] = int(leg["wsum"])
        # the headline's buffers go before the other legs start
        leg_text = leg.pop("text"); leg.pop("d_naf"); leg.pop("naf_buf"); del leg_text
        torch.cuda.synchronize(); ctx.release_scratch(); torch.cuda.empty_cache()
        if args.cfg10_size > 0:
            # ---- BASELINE configs[1] (decode) and [2] (encode) at their own 10 GB, the PMC traffic files' size
            c = one_gpu_leg(ctx, dev, int(args.cfg10_size), args.records, 2024, args.steps, args.warmup)
            cfg10 = {"value": round(c["value"], 3), "ms_per_step": round(c["ms_per_step"], 3), "text_bytes": c["n_text"], "naf_bytes": c["n_naf"],
                     "roundtrip_bit_exact": bool(c["ok"] and c["wsum_ok"]), "ennaf_value": round(c["n_text"] / median(c["enc_times"]) / 1e9, 3),
                     "roofline": decode_roofline(c, c["ms_per_step"])}
            ennaf_roofline = encode_roofline(c)
            extra["cfg10"] = cfg10
            if roofline and cfg10["roofline"]:
                roofline.update({"cfg10_value": cfg10["value"], "cfg10_ms_per_step": cfg10["ms_per_step"], "cfg10_frac": cfg10["roofline"]["frac"],
                                 "cfg10_path_frac": cfg10["roofline"]["path_frac"], "cfg10_roundtrip_bit_exact": cfg10["roundtrip_bit_exact"],
                                 "cfg10_traffic": cfg10["roofline"]["traffic"], "cfg10_ennaf_gbps": cfg10["ennaf_value"]})
            if roofline and ennaf_roofline:
                roofline.update({"cfg10_ennaf_path_frac": ennaf_roofline["path_frac"], "cfg10_ennaf_traffic_ratio": ennaf_roofline.get("call_traffic_ratio")})
            if not args.no_cpu:
                cb = cpu_baseline(c["text"], int(min(args.cpu_sample, c["n_text"])), ctx, e2e_bytes=int(min(args.e2e_size, c["n_text"])))
            del c
            torch.cuda.synchronize(); ctx.release_scratch(); torch.cuda.empty_cache()
    else:
        size = int(args.size) // (world if sharded else 1) if world > 1 else int(args.size)
        spare = 1 << 20                                                    # room behind a slice for what a shard borrows from the next one
        text = synth.fasta_acgt_device(size, n_records=args.records, width=80, seed=2024 + rank, device=dev)
        n_text = int(text.numel())
        text_buf = None
        if sharded:
            # rank r's records are numbered from r * records on: the concatenation of the slices is one FASTA
            text_buf = torch.empty(n_text + spare, dtype=torch.uint8, device=dev)
            text_buf[:n_text] = text
            text = None
            torch.cuda.synchronize(); torch.cuda.empty_cache()        # (the generator's copy goes back to the device: the library's arenas are not torch's)
            text = text_buf[:n_text]
        if n_text <= 40e9:
            ctx.reserve(int(n_text * 1.7) + (2 << 30))
        if not sharded:
            naf_buf = torch.empty(int(n_text * 0.27) + (1 << 20), dtype=torch.uint8, device=dev)
            res_e = [None]
            def enc():
                res_e[0] = ctx.ennaf(text, out=naf_buf)
            enc_times = timed_calls(enc, 5)
            d_naf, rep = res_e[0]
            n_naf = int(d_naf.numel())
            total_text = n_text
            torch.cuda.synchronize(); ctx.release_scratch()
            out = torch.empty(n_text + 64, dtype=torch.uint8, device=dev)
            step = lambda: ctx.unnaf(d_naf, capi.OUT_FASTA, out=out)
            r = step()
            torch.cuda.synchronize()
            ok = equal_in_chunks(r, text)
        else:
            # ---- ONE archive of the whole job's text, made by the sharded encoder; every rank keeps a copy (a quarter of the text)
            opts = shard.make_opts()
            d_naf, rep, sinfo = shard.ennaf_sharded(ctx, text_buf, n_text, opts, dst=0, everywhere=True)
            torch.cuda.synchronize(); dist.barrier()
            enc_times = []
            for _ in range(5):
                torch.cuda.synchronize(); dist.barrier()
                t0 = time.perf_counter()
                shard.ennaf_sharded(ctx, text_buf, n_text, opts, dst=0, everywhere=False)
                torch.cuda.synchronize(); dist.barrier()
                enc_times.append(time.perf_counter() - t0)
            n_naf = int(d_naf.numel())
            torch.cuda.synchronize(); ctx.release_scratch(); torch.cuda.empty_cache()
            total_text = ctx.unnaf_size(d_naf, capi.OUT_FASTA)
            b, e = shard.byte_range(total_text, rank, world)
            out = torch.empty(total_text + 64, dtype=torch.uint8, device=dev) if rank == 0 else None
            scratch = torch.empty(e - b + 64, dtype=torch.uint8, device=dev) if rank != 0 else None
            if world == 1 and scratch is None:                     # --force-sharded on one GPU: the range travels through the communicator to its place (shard._self_exchange)
                scratch = torch.empty(e - b + 64, dtype=torch.uint8, device=dev)
            step = lambda: shard.unnaf_sharded(ctx, d_naf, capi.OUT_FASTA, dst=0, out=out, total=total_text, scratch=scratch, self_exchange=(world == 1))
            r = step()
            torch.cuda.synchronize()
            # full-size check: sum over ranks of the weighted sum of every slice == weighted sum of the gathered text on rank 0, and
            # every rank's own decoded range against the same positions of the text it holds (where they overlap)
            off = n_text * rank                                      # slices are equal-sized by construction
            mine = torch.tensor([weighted_sum(text, off), n_text], dtype=torch.int64, device=dev)
            dist.all_reduce(mine, op=dist.ReduceOp.SUM)
            ok = True
            if rank == 0:
                ok = int(mine[1].item()) == total_text and int(mine[0].item()) == weighted_sum(r, 0) and equal_in_chunks(r[:n_text], text)
            extra["sharded_ennaf"] = {"value": round(total_text / median(enc_times) / 1e9, 3), "unit": "GB/s FASTA in, whole job: split + streams + zstd on every rank, parts gathered to rank 0 (BASELINE configs[4] shape on FASTA)",
                                      "borrowed_bytes": sinfo["halo"], "given_bytes": sinfo["cut"]}
        for _ in range(max(0, args.warmup - 1)):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        dist.barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        if not sharded:
            tot = torch.tensor([float(n_text)], dtype=torch.float64, device=dev)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            total_text = float(tot.item())
        ms_per_step = dt / args.steps * 1e3
        value = float(total_text) * args.steps / dt / 1e9

        if sharded:
            # the two halves of a step on their own: range decode (max over ranks), then the gather of already decoded ranges
            b, e = shard.byte_range(total_text, rank, world)
            buf = out[b:e] if rank == 0 else scratch
            torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                ctx.unnaf_range(d_naf, b, e, capi.OUT_FASTA, out=buf)
            torch.cuda.synchronize()
            td = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(td, op=dist.ReduceOp.MAX)
            dist.barrier()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                shard.gather_ranges(buf[: e - b] if world > 1 else scratch[: e - b], total_text, dst=0, out=out, self_exchange=(world == 1))
            torch.cuda.synchronize(); dist.barrier()
            tg = time.perf_counter() - t0
            # "gather to host" (north_star) without the hop through one GPU: every rank copies its range into its own pinned host buffer
            # (per-GPU D2H, what the C hosts do with NAF_GPUS); step = range decode + the copy, max over ranks
            hsteps = max(1, min(args.steps, 3))
            hbuf = torch.empty(min(e - b, 1 << 30), dtype=torch.uint8).pin_memory()
            def to_host():
                ctx.unnaf_range(d_naf, b, e, capi.OUT_FASTA, out=buf)
                for p in range(0, e - b, int(hbuf.numel())):       # a ring of one pinned buffer: the consumer (a file, a pipe) takes it from there
                    q = min(e - b, p + int(hbuf.numel()))
                    hbuf[: q - p].copy_(buf[p:q], non_blocking=True)
            to_host(); torch.cuda.synchronize(); dist.barrier()
            t0 = time.perf_counter()
            for _ in range(hsteps):
                to_host()
            torch.cuda.synchronize()
            th = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(th, op=dist.ReduceOp.MAX)
            extra["to_host_ms"] = round(float(th.item()) / hsteps * 1e3, 3)
            extra["to_host_value"] = round(float(total_text) * hsteps / float(th.item()) / 1e9, 3)
            del hbuf
            extra["range_decode_ms"] = round(float(td.item()) / args.steps * 1e3, 3)
            extra["gather_ms"] = round(tg / args.steps * 1e3, 3)
            extra["decode_only_value"] = round(float(total_text) * args.steps / float(td.item()) / 1e9, 3)
            if args.fastq_size > 0:
                # configs[4] proper: FASTQ (mixed case, N, full quality range) sharded over the ranks
                if rank == 0:
                    out = None
                scratch = None; buf = None; r = None
                torch.cuda.synchronize(); ctx.release_scratch(); torch.cuda.empty_cache()
                fq = synth.fastq_reads_device(int(args.fastq_size), seed=7 + rank, device=dev)
                fq_buf = torch.empty(fq.numel() + spare, dtype=torch.uint8, device=dev)
                fq_buf[: fq.numel()] = fq
                nfq = int(fq.numel()); del fq
                shard.ennaf_sharded(ctx, fq_buf, nfq, shard.make_opts(), dst=0)
                ts = []
                for _ in range(2):
                    torch.cuda.synchronize(); dist.barrier()
                    t0 = time.perf_counter()
                    fq_naf, fq_rep, _ = shard.ennaf_sharded(ctx, fq_buf, nfq, shard.make_opts(), dst=0)
                    torch.cuda.synchronize(); dist.barrier()
                    ts.append(time.perf_counter() - t0)
                tot = torch.tensor([float(nfq)], dtype=torch.float64, device=dev)
                dist.all_reduce(tot, op=dist.ReduceOp.SUM)
                if rank == 0:
                    back = ctx.unnaf(fq_naf, capi.OUT_FASTQ)
                    # FASTQ comes back with upper-case bases (unnaf.c:442, R3): rank 0's own slice, case-folded, and the total size
                    fq_ok = int(back.numel()) == int(tot.item()) and fastq_same_but_case(back[:nfq], fq_buf[:nfq])
                    extra["sharded_ennaf_fastq"] = {"value": round(float(tot.item()) * len(ts) / sum(ts) / 1e9, 3), "unit": "GB/s FASTQ in (BASELINE configs[4] shape)",
                                                    "text_bytes": int(tot.item()), "naf_ratio": round(fq_naf.numel() / float(tot.item()), 4), "roundtrip_ok_case_folded": fq_ok}
                    del back
                del fq_buf
                if rank == 0:
                    out = torch.empty(e - b + 64, dtype=torch.uint8, device=dev)
                else:
                    scratch = torch.empty(e - b + 64, dtype=torch.uint8, device=dev)

        # ---- per-kernel device time (HIP events on the stream the kernels run on), one extra instrumented step
        ctx.set_timing(True)
        if sharded:
            b, e = shard.byte_range(total_text, rank, world)
            ctx.unnaf_range(d_naf, b, e, capi.OUT_FASTA, out=(out[: e - b + 64] if rank == 0 else scratch))
            my_text = e - b
        else:
            ctx.unnaf(d_naf, capi.OUT_FASTA, out=out)
            my_text = n_text
        legm = {"rep": rep, "n_text": n_text, "n_naf": n_naf, "kt": {n: (ms, k) for n, ms, k in ctx.get_timing()}, "step_streams": ctx.get_timing_streams()}
        ctx.set_timing(False)
        frac_mine = my_text / float(total_text) if sharded else 1.0
        roofline = decode_roofline(legm, ms_per_step if not sharded else extra["range_decode_ms"], my_text=my_text, frac_mine=frac_mine)
        if roofline and sharded:
            # gather-to-root is bound by the root's xGMI ingress, below what one GPU decodes on its own: the other two
            # figures of the step beside `value`, where the driver keeps scalars
            roofline.update({"decode_only_value": extra["decode_only_value"], "to_host_value": extra["to_host_value"], "range_decode_ms": extra["range_decode_ms"],
                             "gather_ms": extra["gather_ms"], "roundtrip_bit_exact": bool(ok),
                             "sharded_ennaf_gbps": extra["sharded_ennaf"]["value"],
                             "sharded_ennaf_fastq_gbps": (extra.get("sharded_ennaf_fastq") or {}).get("value")})

    if rank == 0 and not multi:
        # ---- the other workloads of north_star, on this GPU, in this line (none of them is `value`)
        if args.softmask_size > 0:
            # the mask stream and the toggles of every tile are real work here (they are empty in the headline config)
            sm = synth.softmask_device(synth.fasta_acgt_device(int(args.softmask_size), n_records=24, width=60, seed=7, device=dev))
            extra["softmasked"], _ = side_workload(ctx, sm, capi.OUT_FASTA, "uniform ACGT, 60-column FASTA, 24 records, alternating upper / lower-case runs of 20..600 bases")
            del sm, _
        if args.realistic_size > 0:
            # what an assembled, repeat-masked genome looks like: skewed pair histogram (GC 41 %, CpG depleted), runs of N, IUPAC codes, soft mask
            rg = synth.realistic_genome_device(int(args.realistic_size), device=dev)
            extra["realistic"], rg_naf = side_workload(ctx, rg, capi.OUT_FASTA, "synthetic repeat-masked genome: GC 41 %, CpG at 0.22 of expectation, N runs (telomeres, gaps of 5-100 k), an IUPAC code per Mbase, soft-mask runs of 20..600, 24 records of unequal length, 60-column lines")
            if have_ref() and not args.no_cpu:
                extra["realistic"].update(realistic_vs_reference(ctx, rg, int(min(args.cpu_sample / 4, rg.numel()))))
                if args.ref_full:
                    del rg_naf; rg_naf = None
                    extra["realistic"]["reference_archive_full"] = reference_archive_at_size(ctx, rg, capi.OUT_FASTA)
            del rg, rg_naf
        if args.fastq1_size > 0:
            # BASELINE configs[4] on one GPU: FASTQ, 150-base reads, mixed case + N, full quality range (SURVEY 8(d) cfg5 generator)
            torch.cuda.synchronize(); ctx.release_scratch(); torch.cuda.empty_cache()
            fq = synth.fastq_reads_device(int(args.fastq1_size), seed=7, device=dev)
            ctx.reserve(int(fq.numel() * 1.7) + (2 << 30))
            extra["fastq"], _ = side_workload(ctx, fq, capi.OUT_FASTQ, "FASTQ, 150-base reads `@readN len=150`, ACGT 0.22 each / acgt 0.025 each / N 0.02, quality uniform Phred 0-40 (SURVEY 8(d) cfg5 generator)", fold_case=True)
            # HBM traffic of the leg's two calls, replayed from the round's PMC pass over the same workload (tools/profile_bench.sh: pmc_traffic_fastq.json)
            try:
                pf = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "pmc_traffic_fastq.json")))
                fqb, nafb = int(extra["fastq"]["text_bytes"]), int(extra["fastq"]["naf_bytes"])
                if abs(fqb - 12_500_000_000) < 50_000_000:
                    for side in ("ennaf", "unnaf"):
                        tb = int(pf[side]["call_traffic_bytes"])
                        extra["fastq"][side + "_call_traffic"] = tb
                        extra["fastq"][side + "_traffic_ratio"] = round(tb / (fqb + nafb), 3)
                    extra["fastq"]["traffic_source"] = "profiles/pmc_traffic_fastq.json (replayed)"
            except (OSError, KeyError, ValueError):
                pass
            if have_ref() and not args.no_cpu:
                # the reference's archive of a sample of it decoded here (libzstd's frames: names that copy each other, runs of "len=150")
                extra["fastq"].update(realistic_vs_reference(ctx, fq, int(min(args.cpu_sample / 2, fq.numel())), out_mode=capi.OUT_FASTQ, fold_case=True))
                if args.ref_full:
                    _ = None
                    torch.cuda.synchronize(); ctx.release_scratch(); torch.cuda.empty_cache()
                    extra["fastq"]["reference_archive_full"] = reference_archive_at_size(ctx, fq, capi.OUT_FASTQ, fold_case=True)
            del fq, _
            torch.cuda.synchronize(); ctx.release_scratch(); torch.cuda.empty_cache()
        if args.levels_size > 0 and have_ref() and not args.no_cpu:
            extra["levels"] = levels_leg(ctx, int(args.levels_size), dev)
        if roofline:
            # flat copies of the side figures where the driver's record keeps scalars
            g = lambda k, f: (extra.get(k) or {}).get(f)
            roofline.update({"fastq_unnaf_gbps": g("fastq", "unnaf_value"), "fastq_ennaf_gbps": g("fastq", "ennaf_value"), "fastq_text_bytes": g("fastq", "text_bytes"),
                             "fastq_roundtrip_ok_case_folded": g("fastq", "roundtrip_ok_case_folded"),
                             "fastq_ennaf_traffic_ratio": g("fastq", "ennaf_traffic_ratio"), "fastq_unnaf_traffic_ratio": g("fastq", "unnaf_traffic_ratio"),
                             "realistic_unnaf_gbps": g("realistic", "unnaf_value"), "realistic_ennaf_gbps": g("realistic", "ennaf_value"),
                             "softmasked_unnaf_gbps": g("softmasked", "unnaf_value"), "softmasked_ennaf_gbps": g("softmasked", "ennaf_value")})
            for wl in ("fastq", "realistic"):
                rs = (extra.get(wl) or {}).get("reference_sample") or {}
                ga = rs.get("gpu_unnaf_of_reference_archive") or {}
                full = (extra.get(wl) or {}).get("reference_archive_full") or {}
                # (the figure at the config's own size when the leg ran, the sample's beside it)
                roofline.update({wl + "_ref_archive_gbps": full.get("value", ga.get("value")), wl + "_ref_archive_bit_exact": full.get("bit_exact", ga.get("bit_exact")),
                                 wl + "_ref_archive_text_bytes": full.get("text_bytes", rs.get("text_bytes")),
                                 wl + "_ref_archive_sample_gbps": ga.get("value"), wl + "_ref_archive_sample_bit_exact": ga.get("bit_exact"),
                                 wl + "_ref_archive_ref_unnaf_gbps": rs.get("reference_unnaf_value"),
                                 wl + "_ref_decodes_gpu_archive": rs.get("reference_unnaf_of_gpu_archive_bit_exact")})
            lv = extra.get("levels") or {}
            for k in ("lvl19_ennaf_gbps", "lvl19_ratio_vs_ref", "lvl19_ref_decodes", "lvl19_ref_ennaf_gbps", "long27_ennaf_gbps", "long27_ratio_vs_ref", "long27_ref_decodes", "long27_ref_ennaf_gbps",
                      "long27_ref_archive_unnaf_gbps", "long27_ref_archive_bit_exact", "long27_ref_archive_ref_unnaf_gbps"):
                roofline[k] = lv.get(k)
            ra = (cb or {}).get("gpu_unnaf_of_reference_archive") or {}
            roofline.update({"ref_archive_gbps": ra.get("value"), "ref_archive_bit_exact": ra.get("bit_exact"), "ref_archive_range8_ms": ra.get("range_eighth_ms"),
                             "ref_archive_range8_bit_exact": ra.get("range_eighth_bit_exact"), "ref_archive_text_bytes": ra.get("text_bytes")})
    if rank == 0 and roofline:
        # the scalars that carry the verdict FIRST (the driver's record keeps the head of this object), provenance and detail behind them
        hl = extra.get("headline_ennaf_roofline") or {}
        roofline.setdefault("ennaf_traffic_ratio", hl.get("call_traffic_ratio"))
        roofline.setdefault("ennaf_kernel", hl.get("kernel")); roofline.setdefault("ennaf_kernel_frac", hl.get("frac"))
        head = ["bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "algorithmic_bytes_per_launch", "avg_launch_ms", "path_frac",
                "ennaf_gbps", "ennaf_path_frac", "ennaf_traffic_ratio", "cfg10_value", "cfg10_frac", "cfg10_ennaf_gbps", "fastq_unnaf_gbps", "fastq_ennaf_gbps",
                "ref_archive_gbps", "ref_archive_bit_exact", "fastq_ref_archive_gbps", "fastq_ref_archive_bit_exact", "realistic_unnaf_gbps", "realistic_ennaf_gbps",
                "realistic_ref_archive_gbps", "realistic_ref_archive_bit_exact", "roundtrip_bit_exact", "weighted_sum_equal", "cfg10_roundtrip_bit_exact",
                "fastq_roundtrip_ok_case_folded", "decode_only_value", "to_host_value"]
        roofline = {**{k: roofline[k] for k in head if k in roofline}, **{k: v for k, v in roofline.items() if k not in head}}
    if rank == 0:
        if sharded:
            workload = ("unnaf decode of ONE archive of %.1f GB of synthetic-ACGT FASTA (%.1f GB per GPU, BASELINE configs[3] shape), %d records per GPU, 80-col lines; "
                        "archive made in-run by the sharded GPU ennaf; .naf %d B -> FASTA %d B; every rank holds the archive, rank r emits 1/N of the text, RCCL gather to rank 0"
                        % (total_text / 1e9, n_text / 1e9, args.records, n_naf, total_text))
            par = "one archive, 1/N of the text per GPU by byte range (only the zstd blocks behind it are decoded), gather-to-root as one group of RCCL send/recv"
        elif multi:
            workload = ("unnaf decode of a %.1f GB synthetic-ACGT FASTA archive per GPU, %d records, 80-col lines; archive made in-run by the GPU ennaf; .naf %d B -> FASTA %d B, resident in HBM"
                        % (n_text / 1e9, args.records, n_naf, n_text))
            par = "one archive per GPU, no data-path collective"
        else:
            workload = ("unnaf decode of a %.1f GB synthetic-ACGT FASTA archive on one GPU (the size BASELINE.json's metric is quoted on; configs[1] at 10 GB rides along as cfg10_*), %d records, 80-col lines; "
                        "archive made in-run by the GPU ennaf; .naf %d B -> FASTA %d B, text + archive + decoded text resident in HBM" % (n_text / 1e9, args.records, n_naf, n_text))
            par = "one archive, one GPU, no data-path collective"
        line = {
            "metric": "unnaf GB/s (uncompressed bases out) on 100 GB synthetic FASTA" if abs(float(total_text) - 100e9) < 1e9 else "unnaf GB/s (uncompressed bases out) on synthetic FASTA",
            "value": round(value, 3), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong" if sharded else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": workload, "parallelism": par},
            "roundtrip_bit_exact": bool(ok),
            "ennaf_value": round(float(total_text) / median(enc_times) / 1e9, 3),
            "ennaf_unit": "GB/s FASTA in (device-resident, same data%s; median of %d calls)" % (", sharded over the ranks, parts gathered to rank 0" if sharded else "", len(enc_times)),
            "box": box_id(),
            "naf_ratio": round(n_naf / float(total_text), 4),
            "roofline": roofline, "ennaf_roofline": ennaf_roofline, "cpu_baseline": cb,
        }
        line.update(extra)
    if multi:
        dist.destroy_process_group()
    ctx.close()
    if rank == 0:
        # last thing on stdout (RCCL prints a version banner when the process group goes away)
        sys.stdout.flush(); sys.stderr.flush()
        print(json.dumps(line), flush=True)
        if multi:                                          # ... and another one at exit: nothing of it behind the JSON line
            os.dup2(os.open(os.devnull, os.O_WRONLY), 1)


if __name__ == "__main__":
    main()
