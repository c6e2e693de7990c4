"""GPU tests of the drop-in boundary: the C hosts ennaf/unnaf (naf_amd/bin) run the reference's own
test matrix (tests/golden/ref_tests, ref_cases.json) and interoperate with the real reference binaries."""
import os
import subprocess

import numpy as np

import pytest

from conftest import GOLDEN, ROOT, golden_bytes, naf_cases, ref_cases

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "naf_amd", "bin")


def pipe(text, eargs, uargs):
    e = subprocess.run([os.path.join(BIN, "ennaf"), *eargs, "-c"], input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    u = subprocess.run([os.path.join(BIN, "unnaf"), *uargs, "-c"], input=e.stdout, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    return e, u


@pytest.mark.parametrize("case", ref_cases(), ids=lambda c: c["set"] + "/" + c["name"])
def test_reference_test_matrix_through_the_clis(case):
    text = golden_bytes("ref_tests", case["set"], case["input"])
    eargs = list(case["ennaf_args"])                           # as the reference's suite writes them, `-22 --long 31` of the `large` case included
    e, u = pipe(text, eargs, case["unnaf_args"])
    pre = os.path.join(GOLDEN, "ref_tests", case["set"], case["name"])
    assert e.returncode == 0 and u.returncode == 0, (e.stderr, u.stderr)
    assert u.stdout == open(pre + ".out-ref", "rb").read()
    assert e.stderr == open(pre + ".e.err-ref", "rb").read()
    assert u.stderr == open(pre + ".u.err-ref", "rb").read()


def test_interop_with_the_real_reference(oracle):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    for case in naf_cases():
        naf = golden_bytes("naf", case["name"] + ".naf")
        h = oracle.parse_naf(naf)
        # reference-made archive -> our unnaf CLI == reference unnaf
        mine = subprocess.run([os.path.join(BIN, "unnaf"), "-c"], input=naf, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert mine.returncode == 0 and mine.stdout == oracle.ref_unnaf(naf), case["name"]
        # our ennaf CLI -> reference unnaf == reference ennaf -> reference unnaf
        fq = bool(h.flags & 1)
        text = oracle.ref_unnaf(naf, ("--fastq",) if fq else ("--fasta",))
        args = [a for a in case["ennaf_args"] if not a.startswith("-1") and a not in ("-19", "-3")]
        if "--long" in args:
            i = args.index("--long"); del args[i:i + 2]
        e = subprocess.run([os.path.join(BIN, "ennaf"), *args, "-c"], input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert e.returncode == 0, e.stderr
        assert oracle.ref_unnaf(e.stdout, ("--fastq",) if fq else ("--fasta",)) == text, case["name"]
        if fq:
            continue
        for m in ("--ids", "--names", "--lengths", "--mask", "--total-length", "--number"):
            if m == "--mask" and "--no-mask" in args:
                continue
            a = subprocess.run([os.path.join(BIN, "unnaf"), m, "-c"], input=naf, stdout=subprocess.PIPE, timeout=120).stdout
            assert a == oracle.ref_unnaf(naf, (m,)), (case["name"], m)


def test_clis_on_several_contexts_and_in_ranges(oracle, tmp_path):
    """NAF_GPUS=0,0,0: one host thread and one context per entry (here three on the one device of the test box) -- ennaf cuts the
    file into slices and joins the parts into one archive, unnaf writes every context's share of the text into its place of the
    output file.  NAF_GPU_RANGE_BYTES makes the text leave in small byte ranges, the way a text larger than HBM would."""
    import numpy as np
    from naf_amd import synth
    from test_shard_cpu import check_against_whole
    inputs = {"mixed.fa": synth.fasta_mixed(60, 6000, 60, seed=5), "one.fa": synth.fasta_acgt(700_001, 1, 70, seed=3),
              "reads.fq": synth.fastq_reads(2500, 100, seed=9, var_len=True)}
    for name, text in inputs.items():
        src = tmp_path / name
        src.write_bytes(text)
        fq = name.endswith(".fq")
        for gpus in ("0,0,0", "0,0", "0"):
            env = dict(os.environ, NAF_GPUS=gpus)
            arc = tmp_path / (name + "." + gpus.replace(",", "") + ".naf")
            e = subprocess.run([os.path.join(BIN, "ennaf"), str(src), "-o", str(arc)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=env)
            assert e.returncode == 0 and e.stderr == b"", e.stderr
            naf = arc.read_bytes()
            sp = oracle.split_text(text)
            h = oracle.parse_naf(naf)
            want = [sp.ids, sp.comments, sp.lengths, sp.mask, sp.seq, sp.qual]
            for i in range(6):
                if i == 5 and not fq:
                    continue
                assert oracle.zstd_decompress(h.frame(naf, i), len(want[i]) + 16) == want[i], (name, gpus, i)
            piped = subprocess.run([os.path.join(BIN, "ennaf"), str(src), "-c"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=env)
            assert piped.returncode == 0 and piped.stdout == naf                   # parts in order through a pipe == parts written in place
            expect = oracle.unnaf(naf, -1)
            for rng in (None, "65536"):
                env2 = dict(env)
                if rng:
                    env2["NAF_GPU_RANGE_BYTES"] = rng
                out = tmp_path / "out.txt"
                u = subprocess.run([os.path.join(BIN, "unnaf"), str(arc), "-o", str(out)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=env2)
                assert u.returncode == 0 and u.stderr == b"", u.stderr
                assert out.read_bytes() == expect, (name, gpus, rng)
                u = subprocess.run([os.path.join(BIN, "unnaf"), str(arc), "-c"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=env2)
                assert u.returncode == 0 and u.stdout == expect, (name, gpus, rng)
    # a malformed record in the last slice: the message numbers it from the start of the file
    bad = inputs["reads.fq"].rstrip(b"\n")[:-3] + b"\n"
    (tmp_path / "bad.fq").write_bytes(bad)
    e = subprocess.run([os.path.join(BIN, "ennaf"), str(tmp_path / "bad.fq"), "-c"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=dict(os.environ, NAF_GPUS="0,0,0"))
    with pytest.raises(ValueError) as ex:
        oracle.split_text(bad)
    assert e.returncode == 1 and e.stdout == b"" and e.stderr.decode() == "ennaf error: " + str(ex.value).strip() + "\n"


def test_cli_levels_and_long(oracle):
    """`ennaf -19` and `ennaf -3 --long 27` through the C host: the flags reach the match finder (ennaf.c:247-273, :505), the archives
    stay within 10 % of the real ennaf's with the same flags and come back through both unnafs; on several contexts too."""
    from naf_amd import synth
    for name, text, args in (("repeat_l19", synth.repeat_genome(), ["-19"]),
                             ("repeat_long27", synth.repeat_genome(seed=11, unit=300000, copies=8), ["-3", "--long", "27"])):
        ref_len = os.path.getsize(os.path.join(ROOT, "tests", "golden", "naf", name + ".naf"))
        for env in (dict(os.environ), dict(os.environ, NAF_GPUS="0,0,0")):
            e = subprocess.run([os.path.join(BIN, "ennaf"), *args, "-c"], input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=env)
            assert e.returncode == 0, e.stderr
            naf = e.stdout
            if "NAF_GPUS" not in env:
                assert len(naf) <= 1.10 * ref_len, (name, len(naf), ref_len)
            else:
                assert len(naf) <= 1.35 * ref_len, (name, len(naf), ref_len)      # three shards: each matches inside its own part only
            u = subprocess.run([os.path.join(BIN, "unnaf"), "-c"], input=naf, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
            assert u.returncode == 0 and u.stdout == text
            if oracle.have_ref():
                assert oracle.ref_unnaf(naf) == text


def test_cli_input_in_chunks(oracle, tmp_path):
    """An input larger than the device memory is encoded one chunk after the other (naf_amd/host/ennaf.c: encode_chunked, the shard
    protocol on one device, parts through a temporary file).  Forced here with NAF_GPU_CHUNK_BYTES on inputs of a few MB: the
    archive holds the same six streams as the one-call archive and comes back through both unnafs."""
    from naf_amd import synth
    rng = np.random.default_rng(3)
    fasta = synth.fasta_mixed(40, 60000, 70, 5) + b">last one\n" + bytes(rng.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), 333333)) + b"\n"
    fastq = synth.fastq_reads(30000, 150, seed=2) + synth.fastq_reads(3000, 90, seed=4, var_len=True)
    crlf = fasta[:400000].replace(b"\n", b"\r\n")
    for name, text, sizes in (("fa", fasta, (400000, 1000003, 2500000)), ("fq", fastq, (200000, 3333333)), ("crlf", crlf, (100000,))):
        src = tmp_path / (name + ".txt"); src.write_bytes(text)
        whole = subprocess.run([os.path.join(BIN, "ennaf"), str(src), "-c"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert whole.returncode == 0, whole.stderr
        hw = oracle.parse_naf(whole.stdout)
        for cb in sizes:
            for extra in ([], ["-19"]):
                e = subprocess.run([os.path.join(BIN, "ennaf"), *extra, str(src), "-c"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120,
                                   env=dict(os.environ, NAF_GPU_CHUNK_BYTES=str(cb), NAF_GPU_CLI_TIMING="1"))
                assert e.returncode == 0, e.stderr
                assert (b"ennaf in chunks" in e.stderr) == (len(text) > cb)      # the chunked path did run
                naf = e.stdout
                h = oracle.parse_naf(naf)
                assert naf[: h.header_bytes] == whole.stdout[: hw.header_bytes]
                for i in range(6):
                    if hw.payload_off[i] is None:
                        assert h.payload_off[i] is None
                        continue
                    assert h.orig[i] == hw.orig[i]
                    assert oracle.zstd_decompress(h.frame(naf, i), hw.orig[i] + 64) == oracle.zstd_decompress(hw.frame(whole.stdout, i), hw.orig[i] + 64), (name, cb, i)
                u = subprocess.run([os.path.join(BIN, "unnaf"), "-c"], input=naf, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
                assert u.returncode == 0 and u.stdout == oracle.unnaf(whole.stdout, -1)
                if oracle.have_ref() and len(text) > 3000:
                    assert oracle.ref_unnaf(naf) == u.stdout
    # a record that does not fit a chunk
    e = subprocess.run([os.path.join(BIN, "ennaf"), str(tmp_path / "fa.txt"), "-c"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120,
                       env=dict(os.environ, NAF_GPU_CHUNK_BYTES="4096"))
    assert e.returncode != 0 and b"does not fit a chunk" in e.stderr


def test_line_wrapped_text_takes_the_regular_tile_kernels(tmp_path):
    """A guard on speed paths that parity tests cannot see: nearly every tile of a line-wrapped genome must be judged pure and
    regular (k_enc_count_pure -> k_enc_scatter_regular), soft-masked or not; under NAF_GPU_TRACE=1 the host prints what the library noted."""
    import re
    from naf_amd import synth
    rng = np.random.default_rng(9)
    bases = rng.choice(np.frombuffer(b"ACGTacgtNn", dtype=np.uint8), 3_000_000, p=[.2, .2, .2, .2, .04, .04, .04, .04, .02, .02])
    text = b">chr1 a line-wrapped record\n" + synth.wrap_lines(bases, 70) + b">chr2\n" + synth.wrap_lines(bases[:500_000], 70)
    src = tmp_path / "g.fa"; src.write_bytes(text)
    e = subprocess.run([os.path.join(BIN, "ennaf"), str(src), "-c"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120, env=dict(os.environ, NAF_GPU_TRACE="1"))
    assert e.returncode == 0, e.stderr
    m = re.search(rb"\[reg\] tiles (\d+) need (\d+) regular (\d+)", e.stderr)
    assert m, e.stderr
    tiles, need, regular = (int(x) for x in m.groups())
    assert need <= 8 and regular >= tiles - 12, (tiles, need, regular)
    u = subprocess.run([os.path.join(BIN, "unnaf"), "-c"], input=e.stdout, stdout=subprocess.PIPE, timeout=120)
    assert u.stdout == text


def test_appending_to_a_file_keeps_the_order(tmp_path):
    """`unnaf x.naf >> all.fa` and `ennaf -c in.fa >> out`: a descriptor opened for appending takes pwrite() at the END whatever the
    offset says (Linux), so the threaded positional writer must not be used on it -- the reference's sequential fwrite
    (unnaf/src/output.c:332, ennaf/src/compressor.c:150-173) appends in order.  More than two 16 MiB chunks of text."""
    from naf_amd import synth
    rng = np.random.default_rng(5)
    bases = rng.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), 40_000_000)
    text = b">chr1 forty million bases\n" + synth.wrap_lines(bases, 60)
    src = tmp_path / "a.fa"; src.write_bytes(text)
    naf = tmp_path / "a.naf"
    head = b">already here\nACGT\n"
    for ngpu in ("0", "0,0,0"):
        env = dict(os.environ, NAF_GPUS=ngpu)
        out = tmp_path / ("all_%d.fa" % len(ngpu)); out.write_bytes(head)
        assert subprocess.run([os.path.join(BIN, "ennaf"), str(src), "-o", str(naf)], env=env, timeout=120).returncode == 0
        with open(out, "ab") as f:
            assert subprocess.run([os.path.join(BIN, "unnaf"), str(naf)], stdout=f, env=env, timeout=120).returncode == 0
        assert out.read_bytes() == head + text
        # the archive appended behind other bytes: cut it off again and decode it
        arc = tmp_path / ("arc_%d.bin" % len(ngpu)); arc.write_bytes(b"JUNK")
        with open(arc, "ab") as f:
            assert subprocess.run([os.path.join(BIN, "ennaf"), "-c", str(src)], stdout=f, env=env, timeout=120).returncode == 0
        got = arc.read_bytes()
        assert got[:4] == b"JUNK"
        u = subprocess.run([os.path.join(BIN, "unnaf"), "-c"], input=got[4:], stdout=subprocess.PIPE, timeout=120)
        assert u.returncode == 0 and u.stdout == text


def test_ennaf_from_a_pipe_of_any_size(oracle, tmp_path):
    """process.c:143-150 reads its input 16 KiB at a time and never holds it; here a pipe streams through two pinned buffers into
    the device, and one that outgrows the device buffer (NAF_GPU_PIPE_BYTES, forced small) is spilled to a temporary file and
    encoded chunk by chunk (NAF_GPU_CHUNK_BYTES).  Same six streams as the archive of the file, both unnafs decode it."""
    from naf_amd import synth
    rng = np.random.default_rng(13)
    fasta = synth.fasta_mixed(30, 50000, 70, 9) + b">tail\n" + bytes(rng.choice(np.frombuffer(b"ACGTacgtN", dtype=np.uint8), 200001)) + b"\n"
    fastq = synth.fastq_reads(20000, 150, seed=3)
    for name, text in (("fa", fasta), ("fq", fastq)):
        src = tmp_path / (name + ".txt"); src.write_bytes(text)
        whole = subprocess.run([os.path.join(BIN, "ennaf"), str(src), "-c"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert whole.returncode == 0, whole.stderr
        hw = oracle.parse_naf(whole.stdout)
        want = oracle.unnaf(whole.stdout, -1)
        for env in ({}, {"NAF_GPU_PIPE_BYTES": "300000"}, {"NAF_GPU_PIPE_BYTES": "1000003", "NAF_GPU_CHUNK_BYTES": "700000"}):
            with open(src, "rb") as f:
                cat = subprocess.Popen(["cat"], stdin=f, stdout=subprocess.PIPE)
                e = subprocess.run([os.path.join(BIN, "ennaf"), "-c"] + (["--fastq"] if name == "fq" else []), stdin=cat.stdout, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120,
                                   env=dict(os.environ, NAF_GPU_CLI_TIMING="1", **env))
                cat.wait()
            assert e.returncode == 0, e.stderr
            assert (b"pipe spilled" in e.stderr) == bool(env), e.stderr
            naf = e.stdout
            h = oracle.parse_naf(naf)
            for i in range(6):
                if hw.payload_off[i] is None:
                    assert h.payload_off[i] is None
                    continue
                assert oracle.zstd_decompress(h.frame(naf, i), hw.orig[i] + 64) == oracle.zstd_decompress(hw.frame(whole.stdout, i), hw.orig[i] + 64), (name, env, i)
            u = subprocess.run([os.path.join(BIN, "unnaf"), "-c"], input=naf, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
            assert u.returncode == 0 and u.stdout == want
            if oracle.have_ref():
                assert oracle.ref_unnaf(naf) == want


def test_detached_teardown_is_opt_in_and_dies_with_its_foreground(tmp_path):
    """NAF_GPU_DETACH=1 (host_common.h: detach_teardown): the foreground process leaves when the output is complete, a worker's device
    teardown goes on behind it -- same bytes as the one-process default.  The two are one job until then: a SIGTERM to the pid the
    caller knows reaches the worker (forwarded; PR_SET_PDEATHSIG for a SIGKILL), no orphan finishes the output behind the caller's back."""
    import signal
    import time
    from naf_amd import synth
    text = synth.fasta_acgt(60_000_000, n_records=7, width=70, seed=5)
    src = tmp_path / "d.fa"; src.write_bytes(text)
    one, two = tmp_path / "one.naf", tmp_path / "two.naf"
    assert subprocess.run([os.path.join(BIN, "ennaf"), str(src), "-o", str(one)], timeout=120).returncode == 0
    env = dict(os.environ, NAF_GPU_DETACH="1")
    assert subprocess.run([os.path.join(BIN, "ennaf"), str(src), "-o", str(two)], timeout=120, env=env).returncode == 0
    assert one.read_bytes() == two.read_bytes()
    back = subprocess.run([os.path.join(BIN, "unnaf"), str(two), "-c"], stdout=subprocess.PIPE, timeout=120, env=env)
    assert back.returncode == 0 and back.stdout == text
    time.sleep(1.0)                                            # (the workers of the runs above are gone)
    for sig in (signal.SIGTERM, signal.SIGKILL):
        out = tmp_path / ("killed_%d.naf" % sig)
        p = subprocess.Popen([os.path.join(BIN, "ennaf"), str(src), "-o", str(out)], env=env)
        time.sleep(0.03)                                       # the device is still being opened
        p.send_signal(sig)
        p.wait(timeout=30)
        assert p.returncode == -sig
        time.sleep(2.0)
        assert not out.exists() or out.stat().st_size < len(one.read_bytes()), "a worker finished the output after its foreground process was killed"
