#!/usr/bin/env python3
"""bench.py -- headline benchmark of the NAF hot path on MI355X.

Workload (BASELINE.json configs[1], the config the metric is quoted on for one GPU):
    unnaf decode of a 10 GB synthetic-ACGT .naf, archive resident in HBM, FASTA text produced in HBM.
One "step" = one complete naf_gpu_unnaf pass over the archive (small sections + offset scans + zstd
decode of the sequence stream + 4-bit unpack / mask / line-wrap emit).  The archive is made in-run by
the GPU encoder (naf_gpu_ennaf); its encode throughput is reported as an extra field.

N GPUs (weak scaling): every rank decodes its own 10 GB archive; no data-path collective (the path
shards by independent archives / block ranges); barrier + max-over-ranks timing as the contract asks.

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


def cpu_baseline(text_dev, size_bytes, ctx=None):
    """Reference unnaf/ennaf (oracle/_ref, built from the reference sources) on this box's host cores,
    one thread (the reference is single-threaded), on a bounded sample of the same workload."""
    import numpy as np
    ref_e = os.path.join(ROOT, "oracle", "_ref", "ennaf")
    ref_u = os.path.join(ROOT, "oracle", "_ref", "unnaf")
    if not (os.access(ref_e, os.X_OK) and os.access(ref_u, os.X_OK)):
        return None
    shm = "/dev/shm/naf_bench_%d" % os.getpid()
    os.makedirs(shm, exist_ok=True)
    try:
        sample = text_dev[:size_bytes]
        # cut at a line end so the sample is a well-formed FASTA prefix
        cut = int((sample == 10).nonzero()[-1].item()) + 1
        sample[:cut].cpu().numpy().tofile(os.path.join(shm, "s.fa"))
        env = dict(os.environ, TMPDIR=shm)
        t0 = time.perf_counter()
        subprocess.check_call([ref_e, os.path.join(shm, "s.fa"), "-o", os.path.join(shm, "s.naf")], env=env)
        t_e = time.perf_counter() - t0
        t0 = time.perf_counter()
        subprocess.check_call([ref_u, os.path.join(shm, "s.naf"), "-o", os.path.join(shm, "s.out")])
        t_u = time.perf_counter() - t0
        same = subprocess.call(["cmp", "-s", os.path.join(shm, "s.fa"), os.path.join(shm, "s.out")]) == 0
        out = {"value": round(cut / t_u / 1e9, 4), "unit": "GB/s", "cores": 1, "kind": "reference",
               "sample": "reference unnaf (oracle/_ref, libzstd 1.4.9) on the first %.2f GB of the same FASTA, tmpfs, 1 thread" % (cut / 1e9),
               "ennaf_value": round(cut / t_e / 1e9, 4), "roundtrip_ok": bool(same)}
        # the "whole box" figure -- the reference is single-threaded, so one instance per host core (capped), all
        # decoding the same archive concurrently to /dev/null; aggregate = instances x sample bytes / wall time of the slowest
        ncpu = min(os.cpu_count() or 1, 64)
        if ncpu > 1:
            t0 = time.perf_counter()
            with open(os.devnull, "wb") as dn:
                ps = [subprocess.Popen([ref_u, os.path.join(shm, "s.naf")], stdout=dn) for _ in range(ncpu)]
                rcs = [q.wait() for q in ps]
            t_all = time.perf_counter() - t0
            if all(rc == 0 for rc in rcs):
                out["all_cores"] = {"value": round(ncpu * cut / t_all / 1e9, 3), "unit": "GB/s", "cores": ncpu,
                                    "sample": "%d concurrent reference unnaf instances, each decoding the same %.2f GB sample archive to /dev/null" % (ncpu, cut / 1e9)}
        if ctx is not None:
            # the GPU decoder on the archive the REFERENCE ennaf made of that sample (128 KiB dependent blocks)
            import torch
            from naf_amd import capi
            ref_naf = torch.from_numpy(np.fromfile(os.path.join(shm, "s.naf"), dtype=np.uint8)).to(text_dev.device)
            buf = torch.empty(cut + 64, dtype=torch.uint8, device=text_dev.device)
            r = ctx.unnaf(ref_naf, capi.OUT_FASTA, out=buf)
            torch.cuda.synchronize()
            ok = bool(torch.equal(r, sample[:cut]))
            t0 = time.perf_counter()
            for _ in range(3):
                ctx.unnaf(ref_naf, capi.OUT_FASTA, out=buf)
            torch.cuda.synchronize()
            out["gpu_unnaf_of_reference_archive"] = {"value": round(cut * 3 / (time.perf_counter() - t0) / 1e9, 2), "unit": "GB/s",
                                                      "archive_bytes": int(ref_naf.numel()), "text_bytes": int(cut), "bit_exact": ok}
        return out
    finally:
        subprocess.call(["rm", "-rf", shm])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--size", type=float, default=10e9, help="FASTA bytes per GPU (default: the 10 GB config)")
    ap.add_argument("--records", type=int, default=100)
    ap.add_argument("--cpu-sample", type=float, default=2e9, help="bytes of FASTA timed on the CPU reference")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--shard", action="store_true",
                    help="strong scaling instead of the default weak scaling: ONE archive (same on every rank), every rank decodes its 1/N byte "
                         "range of the text (only the zstd blocks behind it) and the ranges are gathered with one RCCL all_gather "
                         "(BASELINE configs[3] shape); not the default, the driver's contract line is the weak-scaling one")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from naf_amd import capi, synth
    ctx = capi.Context(local)
    size = int(args.size)
    text = synth.fasta_acgt_device(size, n_records=args.records, width=80, seed=2024 + (0 if args.shard else rank), device="cuda:%d" % local)
    n_text = text.numel()
    ctx.reserve(int(n_text * 1.7) + (2 << 30))     # scratch arena sized up front: growth (hipMalloc) and the consolidation after it are not part of a step

    # ---- archive made by the GPU encoder (timed separately; reported as ennaf_value)
    naf_buf = torch.empty(int(n_text * 0.27) + (1 << 20), dtype=torch.uint8, device=text.device)
    torch.cuda.synchronize()
    enc_times = []
    for _ in range(3):
        t0 = time.perf_counter()
        d_naf, rep = ctx.ennaf(text, out=naf_buf)
        torch.cuda.synchronize()
        enc_times.append(time.perf_counter() - t0)
    n_naf = d_naf.numel()
    out = torch.empty(n_text + 64, dtype=torch.uint8, device=text.device)

    if args.shard:
        from naf_amd import shard
        if world == 1:
            step = lambda: ctx.unnaf_range(d_naf, 0, n_text, capi.OUT_FASTA)
        else:
            step = lambda: shard.unnaf_sharded(ctx, d_naf, capi.OUT_FASTA)
    else:
        def step():
            return ctx.unnaf(d_naf, capi.OUT_FASTA, out=out)

    r = step()                                # untimed: bit-exact round trip at full size (size-independent property)
    torch.cuda.synchronize()
    ok = bool(torch.equal(r, text)) if r is not None else True          # --shard: the gathered text lives on rank 0
    for _ in range(max(0, args.warmup - 1)):
        step()
    torch.cuda.synchronize()

    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=text.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        tot = torch.tensor([float(n_text)], dtype=torch.float64, device=text.device)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        total_text = float(n_text) if args.shard else float(tot.item())
    else:
        total_text = float(n_text)
    ms_per_step = dt / args.steps * 1e3
    value = total_text * args.steps / dt / 1e9

    # ---- per-kernel device time (HIP events on the stream the kernels run on), one extra instrumented step
    ctx.set_timing(True)
    ctx.unnaf(d_naf, capi.OUT_FASTA, out=out)
    timing = ctx.get_timing()
    ctx.set_timing(False)
    kt = {n: (ms, k) for n, ms, k in timing}
    packed = (rep.n_bases + 1) // 2
    # algorithmic bytes per launch of each candidate dominant kernel
    alg = {"zstd_huf_literals": rep.section_comp[4] + packed, "unnaf_emit": packed + n_text}
    # Launches on the side contexts' streams are reported as "side:<name>".  The sequence stream's Huffman literals are decoded in a
    # few launches over consecutive block ranges and the text behind a finished range is emitted on a second stream beside the
    # decode of the next one: a kernel's time per step is the sum over its launches of the step -- for the emit,
    # the ones on this context plus "side:unnaf_emit"; "side:zstd_huf_literals" are the side streams' own and not counted.  The
    # launches overlap each other, so each carries the other's contention: the per-kernel fractions are lower bounds.
    def kernel_ms(name):
        ms, k = kt.get(name, (0.0, 0))
        if name == "unnaf_emit":
            ms2, k2 = kt.get("side:unnaf_emit", (0.0, 0)); ms += ms2; k += k2
        return ms, max(k, 1)
    dom = max(alg, key=lambda k: kernel_ms(k)[0])
    dom_ms, dom_launches = kernel_ms(dom)
    achieved = alg[dom] / (dom_ms * 1e-3) / 1e9
    # HBM traffic of that kernel per step, from the committed PMC passes of this same workload (rocprofv3 --pmc
    # FETCH_SIZE / WRITE_SIZE in separate runs, tools/profile_bench.sh; FETCH doubled as the gfx950 guide prescribes)
    traffic, traffic_src = None, None
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        if abs(pm["text_bytes"] - n_text) < 0.01 * n_text and dom in pm["kernels"]:
            traffic = int(pm["kernels"][dom]["fetch_bytes"] + pm["kernels"][dom]["write_bytes"]); traffic_src = pm["source"]
    except (OSError, KeyError, ValueError):
        pass
    roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1), "peak": HBM_PEAK / 1e9, "unit": "GB/s",
                "frac": round(achieved * 1e9 / HBM_PEAK, 4), "traffic": traffic, "traffic_source": traffic_src,
                "algorithmic_bytes_per_launch": int(alg[dom] // dom_launches), "avg_launch_ms": round(dom_ms / dom_launches, 4),
                "kernel_ms_per_step": round(dom_ms, 4), "launches_per_step": dom_launches,
                "path_bytes_per_step": int(n_naf + n_text),
                "path_frac": round((n_naf + n_text) / (ms_per_step * 1e-3) / HBM_PEAK, 4),
                "kernels_ms": {n: round(ms, 3) for n, (ms, k) in sorted(kt.items(), key=lambda x: -x[1][0])[:8]}}

    cb = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cb = cpu_baseline(text, int(min(args.cpu_sample, n_text)), ctx)
    if rank == 0:
        line = {
            "metric": "unnaf GB/s (uncompressed bases out) on synthetic FASTA", "value": round(value, 3), "unit": "GB/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": "strong" if args.shard else "weak", "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "unnaf decode of a %.1f GB synthetic-ACGT FASTA archive per GPU (BASELINE configs[1]), %d records, 80-col lines; archive made in-run by the GPU ennaf; .naf %d B -> FASTA %d B, resident in HBM"
                                   % (n_text / 1e9, args.records, n_naf, n_text),
                       "parallelism": ("one archive, 1/N of the text per GPU by byte range, one RCCL all_gather" if args.shard
                                       else "one archive per GPU, no data-path collective")},
            "roundtrip_bit_exact": ok,
            "ennaf_value": round(n_text / min(enc_times) / 1e9, 3), "ennaf_unit": "GB/s FASTA in (device-resident, same data)",
            "naf_ratio": round(n_naf / n_text, 4),
            "roofline": roofline, "cpu_baseline": cb,
        }
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
