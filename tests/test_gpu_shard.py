"""GPU parity tests of ennaf on several GPUs (include/naf_gpu.h, "ennaf of ONE input on several GPUs"), run as N contexts on the
one device of the test box: the joined archive must hold exactly the six streams the oracle makes of the WHOLE text, its
container framing must equal the reference's, and the real reference unnaf (oracle/_ref) must decode it to the text.  Also the
per-shard records the device reports against the CPU stand-in of tests/shard_standin.py, the reference's die() messages with the
record numbered across shards, and --strict (process.c:98-140) against the real reference's stderr."""
import os
import subprocess

import numpy as np
import pytest

SEED = int(os.environ.get("NAF_TEST_SEED", "0"))          # other texts of the same kinds: NAF_TEST_SEED=n python -m pytest ... (count expectations are seed 0's)

from conftest import ROOT

pytestmark = pytest.mark.gpu
BIN = os.path.join(ROOT, "naf_amd", "bin")


@pytest.fixture(scope="module")
def ctxs():
    import torch
    assert torch.cuda.is_available()
    from naf_amd import capi
    cs = [capi.Context(0) for _ in range(8)]
    yield cs
    for c in cs:
        c.close()


def host(t):
    return t.cpu().numpy().tobytes()


def join(ctxs, text, n, opts=None):
    from naf_amd import shard
    d = ctxs[0].to_device(text)
    naf, rep = shard.ennaf_sharded_local(ctxs[:n], d, opts or shard.make_opts())
    return host(naf), rep


def check(O, ctxs, text, n, seq_type=0, no_mask=False, ref=True):
    from naf_amd import shard
    from test_shard_cpu import check_against_whole
    naf, rep = join(ctxs, text, n, shard.make_opts(seq_type=seq_type, no_mask=no_mask))
    check_against_whole(O, text, naf, rep, seq_type, no_mask)
    sp = O.split_text(text, seq_type, no_mask)
    if sp.n_sequences:
        assert host(ctxs[0].unnaf(ctxs[0].to_device(naf), -1)) == O.unnaf(O.ennaf(text, seq_type, no_mask), -1)   # and the HIP decoder reads it
        if ref and O.have_ref() and len(text) > 3000:             # (the reference's unnaf hangs on very small FASTQ archives)
            fq = sp.format == O.FMT_FASTQ
            args = ("--rna",) if seq_type == 1 else ("--protein",) if seq_type == 2 else ("--text",) if seq_type == 3 else ()
            want = O.ref_unnaf(O.ref_ennaf(text, args + (("--no-mask",) if no_mask else ())))
            assert O.ref_unnaf(naf) == want
    return naf


def test_sharded_fasta_fuzz(ctxs, oracle):
    from test_shard_cpu import fasta_fuzz
    rng = np.random.default_rng(41 + SEED)
    for i in range(40):
        text = fasta_fuzz(rng, int(rng.integers(1, 8)), int(rng.integers(1, 6000)))
        for n in (2, 3, 8):
            check(oracle, ctxs, text, n, ref=(i % 4 == 0))


def test_sharded_one_record_mask_run_across_three_shards(ctxs, oracle):
    seq = b"ACG" + b"acgtn" * 161 + b"TTGCA" * 50 + b"a"
    text = b">chr1 one record\n" + b"".join(seq[a:a + 7] + b"\n" for a in range(0, len(seq), 7))
    for n in (2, 3, 5, 8):
        check(oracle, ctxs, text, n)
    for body in (b"acgt" * 300, b"ACGT" * 300, b"aCgT" * 300, b"a", b"A"):
        text = b">x\n" + b"".join(body[a:a + 11] + b"\n" for a in range(0, len(body), 11))
        for n in (2, 3, 8):
            check(oracle, ctxs, text, n)


def test_sharded_odd_base_counts_at_every_cut(ctxs, oracle):
    """Lines of 7 bases and cuts behind any of them: every shard starts at an odd or even base index as it falls; the packed
    stream must be the one of the whole text, nibble for nibble."""
    rng = np.random.default_rng(8 + SEED)
    bases = np.frombuffer(b"ACGTNacgtRY", dtype=np.uint8)
    for L in (1, 7, 9, 61):
        body = bases[rng.integers(0, len(bases), 4001)].tobytes()
        text = b">r1 d\n" + b"".join(body[a:a + L] + b"\n" for a in range(0, 1501, L)) + b">r2\n" + b"".join(body[a:a + L] + b"\n" for a in range(1501, 4001, L))
        for n in (2, 3, 8):
            check(oracle, ctxs, text, n, ref=(L == 7))


def test_sharded_tiny_and_edge_inputs(ctxs, oracle):
    for text in (b">a\nACGT\n", b">a\n", b">a", b">a b\nAC\n>c\n\n>d\nacgtn", b">x\n" + b"A" * 300, b"\n\n >x\nAC\n".replace(b" ", b""),
                 b">a\r\nAC\r\nGT\r\n>b\r\n\r\nTT", b">x\x01y\nAC\n>z\n"):
        for n in (2, 3, 8):
            check(oracle, ctxs, text, n)


def test_sharded_other_sequence_types(ctxs, oracle):
    from test_shard_cpu import fasta_fuzz
    rng = np.random.default_rng(3 + SEED)
    text = fasta_fuzz(rng, 5, 2500)
    for st, nm in ((oracle.PROTEIN, False), (oracle.TEXT, False), (oracle.DNA, True), (oracle.RNA, False), (oracle.TEXT, True)):
        for n in (2, 3):
            check(oracle, ctxs, text, n, st, nm)


def test_sharded_fastq_mixed_case(ctxs, oracle):
    from naf_amd import synth
    rng = np.random.default_rng(11 + SEED)
    for i in range(10):
        text = synth.fastq_reads(int(rng.integers(1, 400)), int(rng.integers(1, 200)), seed=200 + i, var_len=bool(i % 2))
        for n in (2, 3, 8):
            check(oracle, ctxs, text, n, ref=(i % 3 == 0))
    weird = b"\n\n@r1 c\nAC GT\n+\n!!!!\n\n@r2\tcomment\nACNNxz\n\n+r2 again\n\nIIIIII\n@r3\nA\n+\n~"
    for n in (2, 3):
        check(oracle, ctxs, weird, n, ref=False)          # the reference's unnaf never returns on FASTQ archives this small


def test_sharded_large_roundtrip(ctxs, oracle):
    """200 MB over 8 shards: size-independent property (decode of the joined archive == input) and the same archive sections as
    the one-GPU encoder's (their decoded streams are compared on the device)."""
    import torch
    from naf_amd import synth, shard
    text = synth.fasta_acgt_device(200_000_000, n_records=7, width=80, seed=3)
    naf, rep = shard.ennaf_sharded_local(ctxs, text)
    assert rep.n_sequences == 7 and rep.longest_line == 80
    back = ctxs[0].unnaf(naf, 0)
    assert torch.equal(back, text)
    one, rep1 = ctxs[0].ennaf(text)
    h8, h1 = ctxs[0].parse_header(naf), ctxs[0].parse_header(one)
    assert list(h8.orig_size) == list(h1.orig_size) and h8.n_sequences == h1.n_sequences
    if oracle.have_ref():                                       # the reference reads the first 4 MB of it the same way
        sample = host(ctxs[0].unnaf(naf, 0)[:4_000_000])
        assert oracle.ref_unnaf(host(naf))[:4_000_000] == sample


def test_sharded_dense_mask_units_straight_from_the_case_bits(ctxs, oracle, monkeypatch):
    """A shard's mask units straight from its case bits (k_maskb_units_direct behind k_maskb_census's table of last changes): a text
    whose case changes every few bases -- 3 M changes per 60 MB, far above the 65 536 the path starts at -- over 2 / 3 / 8 shards, with
    runs that cross the cuts and a first base in either case: the joined archive decodes to the text, case and all, and its mask
    stream is the one-GPU encoder's byte for byte; NAF_GPU_MASK_SHORT=1 (the list of positions) gives the same archive."""
    import torch
    from naf_amd import synth, shard
    g = torch.Generator(device="cuda"); g.manual_seed(41)
    base = synth.fasta_acgt_device(60_000_000, n_records=5, width=70, seed=12)
    low = (torch.rand(base.numel(), device="cuda", generator=g) < 0.12) & (base >= 65) & (base <= 90)
    hdr = torch.cumsum((base == 10).to(torch.int32), 0)            # (header lines keep their case: they are the lines that begin with '>')
    is_hdr = torch.zeros_like(low)
    starts = (base == 62).nonzero().flatten().tolist()
    for p0 in starts:
        e = int((base[p0:p0 + 200] == 10).nonzero().flatten()[0].item()) + p0
        is_hdr[p0:e] = True
    text = torch.where(low & ~is_hdr, base + 32, base)
    del low, hdr, is_hdr
    one, _ = ctxs[0].ennaf(text)
    h1 = ctxs[0].parse_header(one)
    m1 = ctxs[0].zstd_decompress(one[h1.payload_off[3]: h1.payload_off[3] + h1.comp_size[3]], int(h1.orig_size[3]) + 64, has_magic=False).clone()
    for n in (2, 3, 8):
        if n == 8: monkeypatch.setenv("NAF_GPU_SHARD_OVERLAP", "1")          # (the sections of a shard side by side on three streams, whatever its size)
        naf, rep = shard.ennaf_sharded_local(ctxs[:n], text)
        if n == 8: monkeypatch.delenv("NAF_GPU_SHARD_OVERLAP")
        assert torch.equal(ctxs[0].unnaf(naf, 0), text), n
        hn = ctxs[0].parse_header(naf)
        mn = ctxs[0].zstd_decompress(naf[hn.payload_off[3]: hn.payload_off[3] + hn.comp_size[3]], int(hn.orig_size[3]) + 64, has_magic=False)
        assert torch.equal(mn, m1), n
        if n == 3:
            monkeypatch.setenv("NAF_GPU_MASK_SHORT", "1")
            naf2, _ = shard.ennaf_sharded_local(ctxs[:n], text)
            monkeypatch.delenv("NAF_GPU_MASK_SHORT")
            assert torch.equal(naf2, naf)


def test_seamed_frames_are_read_in_place_whole_and_by_range(ctxs, monkeypatch, capfd):
    """VERDICT r05 item 4.  The sharded encoder's sequence frame is one run of equal blocks per shard with a short block at every seam (and
    a part is whole blocks + one ragged block, not an even split of two sizes): the decoder's stride index goes on behind the first run
    (zstd_dec.hip: k_runs_next / k_runs_probe) and the frame is read in place like a uniform one -- no block table, no universal
    front -- whole and by byte ranges on 2, 3 and 8 contexts' shares, the same bytes as with NAF_GPU_STRIDE_RUNS=0 and as the text.
    A single shard's frame is uniform outright ([stride] verdict 1)."""
    import torch
    from naf_amd import capi, shard, synth
    text = synth.fasta_acgt_device(96_000_000, n_records=13, width=80, seed=77, device="cuda")
    tb = host(text)
    monkeypatch.setenv("NAF_GPU_DEBUG_STRIDE", "1")
    for n in (1, 2, 3, 8):
        d_naf, rep = shard.ennaf_sharded_local(ctxs[:n], text)
        capfd.readouterr()
        whole = ctxs[0].unnaf(d_naf, capi.OUT_FASTA)
        err = capfd.readouterr().err
        assert torch.equal(whole, text), n
        if n == 1:
            assert "verdict 1" in err and "[uniform?] ok 1 bad 0" in err, err
        else:
            assert "[runs?] ok 1 bad 0" in err, (n, err)
        total = len(tb)
        for parts in (2, 3, 8):
            for r in range(parts):
                b, e = shard.byte_range(total, r, parts)
                got = host(ctxs[r % len(ctxs)].unnaf_range(d_naf, b, e, capi.OUT_FASTA))
                assert got == tb[b:e], (n, parts, r)
        monkeypatch.setenv("NAF_GPU_STRIDE_RUNS", "0")
        assert torch.equal(ctxs[0].unnaf(d_naf, capi.OUT_FASTA), text), n
        monkeypatch.delenv("NAF_GPU_STRIDE_RUNS")
    # an odd count of bases (the stream ends in a Raw block of one byte) and shards of unequal size
    odd = torch.cat([text[:50_000_001], torch.tensor(list(b"A\n"), dtype=torch.uint8, device="cuda")])
    d_naf, _ = shard.ennaf_sharded_local(ctxs[:3], odd)
    assert torch.equal(ctxs[0].unnaf(d_naf, capi.OUT_FASTA), ctxs[0].unnaf(ctxs[0].ennaf(odd)[0], capi.OUT_FASTA))


def test_gather_ranges_of_the_c_abi(ctxs, oracle):
    """naf_gpu_gather_ranges (the product's collective for one process driving N GPUs): every context decodes its byte range of the
    text into a buffer of its own, the ranges are brought together in one buffer of context 0's device -- here the N contexts share the
    one device of the box, so every push is a device-to-device copy; the root's own range is decoded in place and left where it is."""
    import torch
    from naf_amd import synth, shard, capi
    text = synth.softmask_device(synth.fasta_acgt_device(60_000_000, n_records=5, width=60, seed=11))
    naf, _rep = ctxs[0].ennaf(text)
    total = ctxs[0].unnaf_size(naf, capi.OUT_FASTA)
    assert total == text.numel()
    for n in (1, 2, 3, 8):
        out = torch.zeros(total + 64, dtype=torch.uint8, device=text.device)
        parts = []
        for r in range(n):
            b, e = shard.byte_range(total, r, n)
            buf = out[b:e] if r == 0 else torch.empty(e - b + 64, dtype=torch.uint8, device=text.device)
            got = ctxs[r].unnaf_range(naf, b, e, capi.OUT_FASTA, out=buf)
            assert got.numel() == e - b
            parts.append((ctxs[r], got, b))
        ctxs[0].gather_ranges(out, parts)
        torch.cuda.synchronize()
        assert torch.equal(out[:total], text), n


def test_gather_ranges_from_a_c_caller():
    """The same collective with no Python between the caller and the library: tests/c/test_gather.c opens N contexts, decodes unequal
    byte ranges of a golden archive on each, gathers them with naf_gpu_gather_ranges and compares with one whole-text decode -- a FASTA
    archive with a soft mask, a FASTQ one, and the reference's archive of repeats (a frame with matches: the ranges' closures)."""
    exe = os.path.join(ROOT, "tests", "c", "test_gather")
    assert os.access(exe, os.X_OK), "tests/c/test_gather is not built (make)"
    for name, n in (("mixed_60.naf", 3), ("fastq_var.naf", 4), ("repeat_l19.naf", 5), ("acgt_1m2.naf", 8), ("tiny_many.naf", 2)):
        p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "naf", name), str(n)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert p.returncode == 0 and p.stdout.startswith(b"gather ok"), (name, n, p.stdout, p.stderr)


def test_shard_records_match_the_stand_in(ctxs, oracle):
    """naf_gpu_ennaf_shard_begin on the device reports what the CPU stand-in derives from the oracle's view of the same slice."""
    from naf_amd import shard, synth
    from shard_standin import StandInCtx
    from test_shard_cpu import fasta_fuzz
    rng = np.random.default_rng(77 + SEED)
    texts = [fasta_fuzz(rng, 4, 3000) for _ in range(6)] + [synth.fastq_reads(90, 60, seed=9, var_len=True)]
    fields = ("n_sequences", "n_bases", "longest_line", "lead_bases", "n_ids", "n_comments", "n_quality", "mask_changes", "store_mask", "store_quality")
    for text in texts:
        d = ctxs[0].to_device(text)
        opts = shard.make_opts()
        fmt, p0 = ctxs[0].ennaf_sniff(d)
        sfmt, sp0 = StandInCtx(oracle).ennaf_sniff(text)
        assert (fmt, p0) == (sfmt, sp0)
        for n in (2, 3, 8):
            cuts = shard.cuts_local(ctxs[0], d, fmt, p0, n)
            import torch
            scuts = shard.cuts_local(StandInCtx(oracle), torch.frombuffer(bytearray(text), dtype=torch.uint8), fmt, p0, n)
            assert cuts == scuts
            for k in range(n):
                a = ctxs[k].ennaf_shard_begin(d[cuts[k]:cuts[k + 1]], opts, fmt, k, n)
                b = StandInCtx(oracle).ennaf_shard_begin(text[cuts[k]:cuts[k + 1]], opts, fmt, k, n)
                for f in fields:
                    assert getattr(a, f) == getattr(b, f), (f, k, n)
                if a.n_bases:
                    assert (a.first_base, a.last_base) == (b.first_base, b.last_base)
                    if a.mask_changes:
                        assert (a.mask_first_change, a.mask_last_change) == (b.mask_first_change, b.mask_last_change)


def test_sharded_errors_number_records_across_shards(ctxs, oracle):
    from naf_amd import shard, synth
    from naf_amd.capi import NafGpuError
    good = synth.fastq_reads(40, 50, seed=1)
    recs = good.split(b"\n@")
    recs = [recs[0]] + [b"@" + r for r in recs[1:]]
    recs = [r if r.endswith(b"\n") else r + b"\n" for r in recs]
    bad_q = list(recs); bad_q[33] = bad_q[33].rstrip(b"\n")[:-1] + b"\n"              # record 34 loses one quality byte
    bad_plus = list(recs); l = bad_plus[20].split(b"\n"); l[2] = b"x"; bad_plus[20] = b"\n".join(l)
    for broken in (b"".join(bad_q), b"".join(bad_plus), good + b"@last\nACGT\n"):
        with pytest.raises(ValueError) as e0:
            oracle.split_text(broken)
        for n in (2, 3, 8):
            with pytest.raises(NafGpuError) as e1:
                join(ctxs, broken, n)
            assert str(e0.value).strip() in str(e1.value), (n, str(e0.value), str(e1.value))


def _ref_strict(oracle, text, args=()):
    rc, out, err = oracle.ref_ennaf_full(text, ("--strict",) + tuple(args))
    return rc, err.decode("latin1")


def test_strict_matches_the_reference(ctxs, oracle):
    """--strict through the C-ABI (one GPU and sharded) and through the CLI: the reference's message, character and record."""
    from naf_amd import shard
    from naf_amd.capi import NafGpuError
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    fa = b">r1 ok\nACGT\nAC!T\n>r2\x01x c\nAC\n>r3 co\x02mment\nACZT\n"
    cases = [(fa, ()), (b">r1\nACGT\n>r2 c\x7fc\nAC\n", ()), (b">a\nACGT\n>b\nACGT\n>c\nAC.GT\n", ()), (fa, ("--protein",)), (b">a\nAC\x01GT\n", ("--text",)),
             (b">a\nACGU\n", ("--rna",)), (b">a\nACGU\n", ()),
             (b"@r1\nACGT\n+\nIIII\n@r2\nAC.T\n+\nIIII\n", ()), (b"@r1\nACGT\n+\nII\x01I\n@r2\nAC.T\n+\nIIII\n", ()),
             (b"@r1 c\x01\nACGT\n+\nIIII\n", ()), (b"@r1\nACGT\n+\nIIII\n@r\x022\nACGT\n+\nIIII\n", ()),
             (b"@r1\nACGT\n+\nIII\n@r2\nAC.T\n+\nIIII\n", ()),           # the quality length of record 1 comes first
             (b"@r1\nAC.T\n+\nIII\n", ()),                               # the base comes before its record's quality length
             (b">ok\nACGT\n", ()), (b"@ok\nACGT\n+\nIIII\n", ())]
    st = {"--protein": 2, "--text": 3, "--rna": 1}
    for text, args in cases:
        rc, err = _ref_strict(oracle, text, args)
        seq_type = st.get(args[0], 0) if args else 0
        for n in (1, 2, 3):
            opts = shard.make_opts(seq_type=seq_type, strict=True)
            if rc == 0:
                join(ctxs, text, n, opts)
                continue
            with pytest.raises(NafGpuError) as e:
                join(ctxs, text, n, opts)
            assert err.startswith("ennaf error: ") and err[len("ennaf error: "):].strip() in str(e.value), (text, n, err, str(e.value))
        p = subprocess.run([os.path.join(BIN, "ennaf"), "--strict", *args, "-c"], input=text, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
        assert (p.returncode, p.stderr.decode("latin1")) == (rc, err), (text, args)
    p = subprocess.run([os.path.join(BIN, "ennaf"), "--strict", "--well-formed", "-c"], input=b">a\nAC\n", stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert p.returncode == 1 and p.stderr == b"ennaf error: '--well-formed' and '--strict' can't be used together\n"


def test_sharded_ennaf_at_a_high_level(ctxs, oracle):
    """Shards of one archive at level 19: every shard matches inside its own part of a stream; the joined frames decode."""
    from naf_amd import synth, shard
    text = synth.repeat_genome(seed=5, unit=30000, copies=20)
    gpu = ctxs[0]
    naf, rep = shard.ennaf_sharded_local(ctxs[:3], gpu.to_device(text), shard.make_opts(level=19))
    mine = host(naf)
    want = oracle.unnaf(oracle.ennaf(text), -1)
    assert oracle.unnaf(mine, -1) == want
    assert host(gpu.unnaf(gpu.to_device(mine), -1)) == want
    if oracle.have_ref():
        assert oracle.ref_unnaf(mine) == want
    os.environ["NAF_GPU_PROBE"] = "0"                               # level 1 without its look at the stream: entropy coding only
    try:
        assert len(mine) < 0.5 * len(host(shard.ennaf_sharded_local(ctxs[:3], gpu.to_device(text), shard.make_opts())[0]))
    finally:
        del os.environ["NAF_GPU_PROBE"]
    lvl1 = host(shard.ennaf_sharded_local(ctxs[:3], gpu.to_device(text), shard.make_opts())[0])      # and with it: the repeats are found
    assert oracle.unnaf(lvl1, -1) == want and len(lvl1) < 1.5 * len(mine)


def test_window_descriptor_of_a_sharded_frame_follows_the_options(ctxs, oracle):
    """The frame header of every stream is written by the FIRST shard; at level >= 2 / --long the later shards match across blocks
    inside the window of the level, so the header must announce that window even when the first shard's part of a stream is a few
    bytes (one chromosome: an ids stream of 5 bytes) -- the reference's streaming decoder sizes its history from that field
    (unnaf/src/input.c:262-285)."""
    from naf_amd import shard
    rng = np.random.default_rng(77 + SEED)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    big = acgt[rng.integers(0, 4, 1300000)].tobytes()
    unit = acgt[rng.integers(0, 4, 3000)].tobytes()
    text = b">chr1\n" + b"".join(big[a:a + 80] + b"\n" for a in range(0, len(big), 80))
    for k in range(400):                                             # many records whose ids / names / bases repeat far back
        u = unit[(k * 7) % 100:]
        text += b">scaffold_%06d some repeated description of a contig\n" % k + b"".join(u[a:a + 80] + b"\n" for a in range(0, len(u), 80))
    for level, long_log, want in ((3, 0, 21), (1, 27, None), (19, 0, 23)):
        opts = shard.make_opts(level=level, long_log=long_log)
        naf, rep = join(ctxs, text, 3, opts)
        h = oracle.parse_naf(naf)
        for i in range(5):
            fr = h.frame(naf, i)                                     # with the magic number in front
            assert fr[4] & 0x20 == 0                                 # not single-segment: a Window_Descriptor follows
            wlog = 10 + (fr[5] >> 3)
            if want is not None:
                assert wlog == want, (level, i, wlog)
            elif i == 4:
                assert wlog == long_log
        assert host(ctxs[0].unnaf(ctxs[0].to_device(naf), -1)) == text
        if oracle.have_ref():
            assert oracle.ref_unnaf(naf) == text


def _two_rank_worker(rank, world, port, q, kind):
    """One process of a two-rank job on the ONE device of the test box: the real library (capi.Context(0)) on either side of real
    process boundaries, torch.distributed over gloo (RCCL needs a device per rank; naf_amd/shard.py stages device tensors through
    the host for gloo) -- sharded encode into one archive on every rank, then the sharded decode of it gathered to rank 0."""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")                  # the box's host name may not resolve
    import datetime
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=120))
    try:
        from naf_amd import capi, shard as sh, synth
        from oracle import oracle as O
        import test_shard_cpu as me
        torch.cuda.set_device(0)
        ctx = capi.Context(0)
        rng = np.random.default_rng(123 + SEED)
        if kind == "fasta":
            text = b"\n \n" + me.fasta_fuzz(rng, 4, 40000, width=60) + synth.fasta_acgt(600_001, 2, 80, seed=5)
        elif kind == "fastq":
            text = synth.fastq_reads(4000, 120, seed=6, var_len=True)
        else:                                                          # level 3: matches across blocks inside every shard's part
            text = synth.repeat_genome(seed=8, unit=30000, copies=12)
        a = [0, len(text) * 2 // 5, len(text)]                          # uneven nominal slices, cut anywhere
        mine = text[a[rank]:a[rank + 1]]
        buf = torch.zeros(len(mine) + (1 << 16), dtype=torch.uint8, device="cuda")
        buf[: len(mine)] = torch.frombuffer(bytearray(mine), dtype=torch.uint8).cuda()
        opts = sh.make_opts(level=3 if kind == "repeat" else 1)
        naf, rep, extra = sh.ennaf_sharded(ctx, buf, len(mine), opts, dst=0, everywhere=True)
        arc = naf.cpu().numpy().tobytes()
        ok = True
        try:
            if kind != "repeat":
                me.check_against_whole(O, text, arc, rep)
            mode = capi.OUT_FASTQ if kind == "fastq" else capi.OUT_FASTA
            want = O.unnaf(O.ennaf(text), mode)
            got = sh.unnaf_sharded(ctx, naf, mode, dst=0)
            if rank == 0:
                assert got.cpu().numpy().tobytes() == want
                if O.have_ref() and kind != "fastq":
                    assert O.ref_unnaf(arc) == want
            else:
                assert got is None
            # "gather to host" without a hop through one GPU (what the C hosts do under NAF_GPUS): every rank decodes its byte range and
            # writes it into its place of ONE file over its own link (naf_gpu_write_file); rank 0 reads the file back
            total = ctx.unnaf_size(naf, mode)
            b0, e0 = sh.byte_range(total, rank, world)
            piece = ctx.unnaf_range(naf, b0, e0, mode)
            path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "naf_to_host_%d.out" % port)
            if rank == 0:
                with open(path, "wb") as f:
                    f.truncate(total)
            dist.barrier()
            fd = os.open(path, os.O_WRONLY)
            if e0 > b0:
                ctx.write_file(fd, b0, piece)
            os.close(fd)
            dist.barrier()
            if rank == 0:
                with open(path, "rb") as f:
                    assert f.read() == want
                os.remove(path)
        except AssertionError as e:
            ok = "rank %d: %r" % (rank, e)
        ctx.close()
        q.put((rank, ok, extra))
    except Exception as e:                                             # noqa: BLE001 -- the parent must hear about it
        q.put((rank, "rank %d: %r" % (rank, e), {}))
    dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["fasta", "fastq", "repeat"])
def test_two_processes_one_device_real_library_over_gloo(kind):
    """the shard protocol and the gather across REAL process boundaries with the HIP library on both sides."""
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = 33500 + os.getpid() % 2000 + {"fasta": 0, "fastq": 11, "repeat": 23}[kind]
    ps = [mpc.Process(target=_two_rank_worker, args=(r, 2, port, q, kind), daemon=True) for r in range(2)]
    for p in ps:
        p.start()
    try:
        res = sorted(q.get(timeout=240) for _ in ps)
    finally:
        for p in ps:
            p.join(30)
            if p.is_alive():
                p.kill()
    assert all(r[1] is True for r in res), res
    assert res[1][2]["cut"] > 0 and res[0][2]["halo"] == res[1][2]["cut"]


_RCCL_SELF = r"""
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = %(port)r
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from naf_amd import capi, shard, synth
ctx = capi.Context(0)
text = synth.fasta_acgt_device(40_000_000, n_records=9, width=80, seed=3, device="cuda")
d_naf, _ = ctx.ennaf(text)
want = ctx.unnaf(d_naf, capi.OUT_FASTA).clone()
assert dist.get_backend() == "nccl"
got = shard.unnaf_sharded(ctx, d_naf, capi.OUT_FASTA, self_exchange=True)
torch.cuda.synchronize()
assert torch.equal(got, want) and torch.equal(got, text), "unnaf_sharded through the communicator"
piece = want[: want.numel()].clone()
g2 = shard.gather_ranges(piece, int(piece.numel()), self_exchange=True)
torch.cuda.synchronize()
assert torch.equal(g2, want), "gather_ranges through the communicator"
outs = [torch.empty_like(piece)]
shard._all_gather(outs, piece, None)
torch.cuda.synchronize()
assert torch.equal(outs[0], want), "all_gather of one rank"
# a range of more than a GiB (a rank's share of the headline text is 12.5 GB): RCCL 2.26's exchange of a rank with itself loses the
# second half of such a message as ONE transfer (profiles/r06_rccl_self_exchange.txt); shard._pieces posts it 512 MiB at a time
big = torch.arange(0, 1_400_000_000 // 8, dtype=torch.int64, device="cuda").view(torch.uint8)
g3 = shard.gather_ranges(big, int(big.numel()), self_exchange=True)
torch.cuda.synchronize()
assert torch.equal(g3, big), "gather_ranges of 1.4 GB through the communicator"
del g3, big
ctx.close()
dist.destroy_process_group()
print("rccl self exchange ok")
"""


def test_rccl_branch_of_the_gather_on_one_rank():
    """`shard.gather_ranges` / `unnaf_sharded` have an RCCL branch (`_p2p_post` with device tensors over the
    `nccl` backend) that no one-GPU box reached -- a world of one rank returned before it.  Under `self_exchange` the rank posts the
    receive and the send of its own range to itself as ONE group of point-to-point transfers, so the communicator moves 40 MB of decoded
    text on this box; a one-rank all_gather goes the same way.  A process of its own under a timeout: a communicator that hangs must
    not take the suite (or the box) with it."""
    import subprocess, sys
    src = _RCCL_SELF % {"root": ROOT, "port": str(34100 + os.getpid() % 1500)}
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0 and "rccl self exchange ok" in r.stdout, (r.returncode, r.stdout[-800:], r.stderr[-1500:])
