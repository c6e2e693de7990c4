"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C-ABI, against
the oracle and the golden vectors made by the real reference.  Nothing here reads /root/reference."""
import hashlib
import os

import numpy as np
import pytest

SEED = int(os.environ.get("NAF_TEST_SEED", "0"))          # other texts of the same kinds: NAF_TEST_SEED=n python -m pytest ... (count expectations are seed 0's)

from conftest import golden_bytes, naf_cases, zstd_cases

pytestmark = pytest.mark.gpu


def sha(b):
    return hashlib.sha256(b).hexdigest()


@pytest.fixture(scope="module")
def gpu():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from naf_amd import capi
    ctx = capi.Context(0)
    yield ctx
    ctx.close()


def host(t):
    return t.cpu().numpy().tobytes()


@pytest.mark.parametrize("executor", ["auto", "hbm", "hbm-smallcollapse", "hbm-nearcollapse", "hbm-nocollapse", "batch", "serial"])
@pytest.mark.parametrize("case", zstd_cases(), ids=lambda c: c["name"])
def test_zstd_decode_golden_frames(gpu, case, executor, monkeypatch):
    """libzstd-made frames (every level, long windows, multi-threaded, streaming): frames whose blocks regenerate at most 16 KiB
    run their sequences in the LDS executor, the others as dataflow (k_lz_prep / _deps / _exec); "hbm" forces the latter for every
    frame, "hbm-nocollapse" without the move of sources back along chains of copies (k_lz_collapse), "batch" and "serial" the two block-ordered executors kept as
    cross-checks."""
    monkeypatch.setenv("NAF_GPU_EXEC_LDS", "1" if executor == "auto" else "0")
    if executor == "hbm-nocollapse":
        monkeypatch.setenv("NAF_GPU_EXEC_COLLAPSE", "0")
    if executor == "hbm-nearcollapse":                        # within units only (k_lz_collapse without k_lz_collapse_far)
        monkeypatch.setenv("NAF_GPU_EXEC_COLLAPSE", "n")
    if executor == "hbm-smallcollapse":                       # units of 1024 sequences by single wavefronts whatever the frame
        monkeypatch.setenv("NAF_GPU_EXEC_COLLAPSE", "s")
    if executor in ("batch", "serial"):
        monkeypatch.setenv("NAF_GPU_EXEC", executor)
    frame = golden_bytes("zstd", case["name"] + ".zst")
    out = gpu.zstd_decompress(gpu.to_device(frame), case["len"] + 64)
    got = host(out)
    assert len(got) == case["len"] and sha(got) == case["sha256"]


@pytest.mark.parametrize("lever", [("NAF_GPU_SEQ_WAVE", "l"), ("NAF_GPU_SEQ_WAVE", "all"), ("NAF_GPU_SEQ_REP", "walk")], ids=lambda l: l[0][8:] + "=" + l[1])
@pytest.mark.parametrize("case", zstd_cases(), ids=lambda c: c["name"])
def test_zstd_decode_golden_frames_under_the_sequence_walks_kept_as_cross_checks(gpu, case, lever, monkeypatch):
    """The sequences of a block under tables of its own are walked by k_decode_seq_wave2 (a branch-free chain over the state bits, a lane
    per sequence for the values, the repeat offsets as a prefix scan of the sequences' turns).  Kept beside it: the single-lane walk
    (SEQ_WAVE=l), the same kernel for blocks under the predefined tables too (SEQ_WAVE=all), the repeat offsets one sequence after the
    other (SEQ_REP=walk) -- every golden frame must come out the same under each."""
    monkeypatch.setenv(*lever)
    frame = golden_bytes("zstd", case["name"] + ".zst")
    got = host(gpu.zstd_decompress(gpu.to_device(frame), case["len"] + 64))
    assert len(got) == case["len"] and sha(got) == case["sha256"]


def test_zstd_decode_matches_oracle_on_naf_sections(gpu, oracle):
    for case in naf_cases():
        naf = golden_bytes("naf", case["name"] + ".naf")
        h = oracle.parse_naf(naf)
        for i in range(6):
            if h.payload_off[i] is None:
                continue
            ref = oracle.zstd_decompress(h.frame(naf, i))
            sec = naf[h.payload_off[i]: h.payload_off[i] + h.comp[i]]
            out = gpu.zstd_decompress(gpu.to_device(sec), len(ref) + 64, has_magic=False)
            assert host(out) == ref, (case["name"], i)


def _hand_frame(blocks):
    """zstd frame (magic, single-segment off, window 2^23, no checksum) from (type, payload, regen) blocks: type 0 raw, 1 RLE"""
    out = bytearray(b"\x28\xb5\x2f\xfd\x00\x68")
    for i, (t, payload, regen) in enumerate(blocks):
        h = (1 if i + 1 == len(blocks) else 0) | (t << 1) | (regen << 3)
        out += bytes([h & 0xFF, (h >> 8) & 0xFF, (h >> 16) & 0xFF]) + payload
    return bytes(out)


@pytest.mark.parametrize("total", [6 << 10, 60 << 10, 300 << 10, 9 << 20])
def test_zstd_block_index_survives_lookalike_headers(gpu, total):
    """The parallel block index tests bytes as candidate block headers (16 KiB chunks with every byte tested for frames of up to
    4 MiB, 1 MiB chunks above).  Payloads full of bytes that READ as valid header chains -- runs of zeros (empty raw blocks), of
    0x02 (RLE blocks, a hop of 4), tiny raw blocks -- and real empty raw blocks in the middle of the frame must not change
    the result: candidates only decide how much of the speculation is reused."""
    rng = np.random.default_rng(total)
    blocks, expect = [], bytearray()
    while len(expect) < total:
        kind = int(rng.integers(0, 6))
        if kind == 0:                                             # raw block of zeros: 3-byte groups read as empty raw blocks
            n = int(rng.integers(30, 5000)); blocks.append((0, bytes(n), n)); expect += bytes(n)
        elif kind == 1:                                           # raw block of 02 00 00 xx: each group reads as an RLE block
            n = int(rng.integers(8, 3000)) * 4; pl = bytes([2, 0, 0, 7] * (n // 4)); blocks.append((0, pl, n)); expect += pl
        elif kind == 2:                                           # a real empty raw block in the middle of the frame
            blocks.append((0, b"", 0))
        elif kind == 3:                                           # RLE block
            n = int(rng.integers(1, 100000)); b = int(rng.integers(0, 256)); blocks.append((1, bytes([b]), n)); expect += bytes([b]) * n
        elif kind == 4:                                           # a burst of tiny raw blocks
            for _ in range(int(rng.integers(1, 400))):
                n = int(rng.integers(1, 6)); pl = rng.integers(0, 256, n, dtype=np.uint8).tobytes(); blocks.append((0, pl, n)); expect += pl
        else:                                                     # random raw data up to the block size limit
            n = int(rng.integers(1, 131072)); pl = rng.integers(0, 256, n, dtype=np.uint8).tobytes(); blocks.append((0, pl, n)); expect += pl
    frame = _hand_frame(blocks)
    out = gpu.zstd_decompress(gpu.to_device(frame), len(expect) + 64)
    got = host(out)
    assert len(got) == len(expect) and sha(got) == sha(bytes(expect))


def test_zstd_rejects_corrupt_frames(gpu):
    from naf_amd.capi import NafGpuError
    frame = bytearray(golden_bytes("zstd", "ids_l3.zst"))
    with pytest.raises(NafGpuError):
        gpu.zstd_decompress(gpu.to_device(bytes(frame[: len(frame) // 2])), 1 << 20)
    frame[3] ^= 0xFF
    with pytest.raises(NafGpuError):
        gpu.zstd_decompress(gpu.to_device(bytes(frame)), 1 << 20)


MODES = {"fasta": (0, True, -1), "seq": (2, True, -1), "sequences": (3, True, -1), "4bit": (4, True, -1),
         "fasta_nomask": (0, False, -1), "fasta_ll13": (0, True, 13), "fasta_ll0": (0, True, 0), "fastq": (1, True, -1)}


@pytest.mark.parametrize("path", ["fused", "long", "span", "short", "slow"])
@pytest.mark.parametrize("case", naf_cases(), ids=lambda c: c["name"])
def test_unnaf_matches_reference_outputs(gpu, case, path, monkeypatch):
    """Bit-exact against the outputs of the real reference unnaf on reference-made archives, through each of the
    emit paths: fused decode+emit (literal-only frames), decode then the tile-indexed long-record kernel, the older
    span kernel, the short-record segment-composing kernel, per-byte emit."""
    monkeypatch.setenv("NAF_GPU_FORCE_SLOW", "1" if path == "slow" else "0")
    monkeypatch.setenv("NAF_GPU_EMIT", path if path in ("long", "span", "short") else "")
    monkeypatch.setenv("NAF_GPU_FUSE", "1" if path == "fused" else "0")
    naf = golden_bytes("naf", case["name"] + ".naf")
    d = gpu.to_device(naf)
    for m, (mode, use_mask, ll) in MODES.items():
        if m not in case["outputs"]:
            continue
        exp = case["outputs"][m]
        assert gpu.unnaf_size(d, mode, use_mask, ll) == exp["len"], m
        got = host(gpu.unnaf(d, mode, use_mask, ll))
        assert len(got) == exp["len"], m
        assert sha(got) == exp["sha256"], m


@pytest.mark.parametrize("case", naf_cases(), ids=lambda c: c["name"])
def test_unnaf_record_tables_the_long_way(gpu, case, monkeypatch):
    """The record tables of an archive of few records come from one launch behind the small-frame decoder (k_side_tables); with
    NAF_GPU_SIDE_FUSED=0 they come from the chain of kernels archives of many records take.  Same reference outputs."""
    monkeypatch.setenv("NAF_GPU_SIDE_FUSED", "0")
    d = gpu.to_device(golden_bytes("naf", case["name"] + ".naf"))
    for m, (mode, use_mask, ll) in MODES.items():
        if m not in case["outputs"]:
            continue
        exp = case["outputs"][m]
        got = host(gpu.unnaf(d, mode, use_mask, ll))
        assert len(got) == exp["len"] and sha(got) == exp["sha256"], m


def test_histogram_matches_numpy(gpu):
    rng = np.random.default_rng(2 + SEED)
    for n in (0, 1, 7, 65536, 65537, 1000003):
        d = rng.integers(0, 256, n, dtype=np.uint8)
        d[: n // 3] = 65
        got = gpu.histogram(gpu.to_device(d.tobytes())) if n else [0] * 256
        assert got == np.bincount(d, minlength=256).tolist(), n


def test_unnaf_survives_corrupt_sections(gpu, oracle):
    """A damaged section must come back as an error (or, when the damage happens to decode, as some text) -- never a crash or a
    hang; the side streams are decoded by helper host threads, so this also walks their error paths."""
    from naf_amd.capi import NafGpuError
    rng = np.random.default_rng(8 + SEED)
    for name in ("mixed_60", "fastq_4k"):
        naf = bytearray(golden_bytes("naf", name + ".naf"))
        h = oracle.parse_naf(bytes(naf))
        for i in range(6):
            if h.payload_off[i] is None or h.comp[i] < 8:
                continue
            for trial in range(6):
                bad = bytearray(naf)
                pos = h.payload_off[i] + int(rng.integers(0, h.comp[i]))
                bad[pos] ^= 1 << int(rng.integers(0, 8))
                if trial == 5:
                    bad = bad[: h.payload_off[i] + h.comp[i] // 2]                # truncated inside the section
                for mode in (-1, 2):
                    try:
                        gpu.unnaf(gpu.to_device(bytes(bad)), mode)
                    except NafGpuError:
                        pass


def test_small_frames_damaged_the_oracle_is_the_judge(gpu, oracle):
    """The small-frame decoder (one wavefront per frame, zstd_dec.hip small_frame_decode) on the reference-made section frames of the
    golden archives with one bit flipped, a byte replaced, or the frame cut short: the oracle's verdict -- these bytes, or an error --
    must be this decoder's; and whole archives damaged in their small sections end the same way with the record tables made in one
    launch (k_side_tables) as with NAF_GPU_SIDE_FUSED=0."""
    import os
    from naf_amd.capi import NafGpuError
    rng = np.random.default_rng(77 + SEED)
    n_frames = n_ok = n_err = 0
    for case in naf_cases():
        naf = golden_bytes("naf", case["name"] + ".naf")
        h = oracle.parse_naf(naf)
        for i in range(4):
            if h.payload_off[i] is None or not (8 <= h.comp[i] <= 16384) or h.orig[i] > 32768:
                continue
            fr = bytes(naf[h.payload_off[i]:h.payload_off[i] + h.comp[i]])
            n_frames += 1
            for trial in range(24):
                bad = bytearray(fr)
                kind = trial % 3
                if kind == 0: bad[int(rng.integers(0, len(bad)))] ^= 1 << int(rng.integers(0, 8))
                elif kind == 1: bad[int(rng.integers(0, len(bad)))] = int(rng.integers(0, 256))
                else: bad = bad[: int(rng.integers(1, len(bad)))]
                try:
                    want = oracle.zstd_decompress(b"\x28\xb5\x2f\xfd" + bytes(bad), h.orig[i] + 64)
                except ValueError:
                    want = None
                try:
                    got = host(gpu.zstd_decompress(gpu.to_device(bytes(bad)), h.orig[i] + 64, has_magic=False))
                except NafGpuError:
                    got = None
                assert got == want, (case["name"], i, trial, None if got is None else len(got), None if want is None else len(want))
                n_ok += want is not None; n_err += want is None
    assert n_frames >= 6 and n_ok and n_err
    for name in ("mixed_60", "fastq_4k"):
        naf = bytearray(golden_bytes("naf", name + ".naf"))
        h = oracle.parse_naf(bytes(naf))
        for i in range(3):
            if h.payload_off[i] is None or h.comp[i] < 8:
                continue
            for trial in range(10):
                bad = bytearray(naf)
                bad[h.payload_off[i] + int(rng.integers(0, h.comp[i]))] ^= 1 << int(rng.integers(0, 8))
                res = []
                for fused in ("1", "0"):
                    os.environ["NAF_GPU_SIDE_FUSED"] = fused
                    try:
                        res.append(("ok", sha(host(gpu.unnaf(gpu.to_device(bytes(bad)), -1)))))
                    except NafGpuError as e:
                        res.append(("error", e.code, str(e)))
                os.environ.pop("NAF_GPU_SIDE_FUSED")
                assert res[0] == res[1], (name, i, trial, res)


def test_unnaf_reference_suite(gpu, oracle):
    """The reference's own tests: oracle-made archive -> GPU unnaf == *.out-ref."""
    from conftest import ref_cases
    for case in ref_cases():
        ea, ua = case["ennaf_args"], case["unnaf_args"]
        if "--charcount" in ua:
            continue
        st = oracle.RNA if "--rna" in ea else oracle.PROTEIN if "--protein" in ea else oracle.TEXT if "--text" in ea else oracle.DNA
        text = golden_bytes("ref_tests", case["set"], case["input"])
        naf = oracle.ennaf(text, st, "--no-mask" in ea)
        mode = 2 if "--seq" in ua else 3 if "--sequences" in ua else -1
        exp = golden_bytes("ref_tests", case["set"], case["name"] + ".out-ref")
        got = host(gpu.unnaf(gpu.to_device(naf), mode, "--no-mask" not in ua)) if len(naf) else b""
        assert got == exp, case["name"]


def test_unnaf_range_sharding_is_consistent(gpu):
    """Per-GPU byte ranges (multi-GPU sharding) concatenate to the whole text."""
    naf = golden_bytes("naf", "mixed_60.naf")
    d = gpu.to_device(naf)
    whole = host(gpu.unnaf(d, 0))
    n = len(whole)
    cuts = [0, 1, 17, n // 3, n // 3 + 5, 2 * n // 3, n - 1, n]
    parts = [host(gpu.unnaf_range(d, a, b, 0)) for a, b in zip(cuts[:-1], cuts[1:])]
    assert b"".join(parts) == whole


@pytest.mark.parametrize("emit", ["long", "span", "short"])
def test_unnaf_range_on_own_archives_decodes_only_needed_blocks(gpu, oracle, emit, monkeypatch):
    """Own archives (independent blocks) take the block-range path: every cut of FASTA / FASTQ / --seq /
    --sequences text equals the slice of the whole text."""
    from naf_amd import synth
    monkeypatch.setenv("NAF_GPU_EMIT", emit)
    rng = np.random.default_rng(4 + SEED)
    texts = [synth.fasta_acgt(700000, 5, 80, seed=21), synth.fasta_mixed(25, 30000, 60, seed=22), synth.fastq_reads(3000, 150, seed=23)]
    for text in texts:
        d_naf, rep = gpu.ennaf(gpu.to_device(text))
        for mode in (-1, 0, 2, 3):
            if mode == 0 and text[:1] == b"@":
                pass
            whole = host(gpu.unnaf(d_naf, mode))
            n = len(whole)
            cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, 6)]))
            for a, b in zip(cuts[:-1], cuts[1:]):
                assert host(gpu.unnaf_range(d_naf, a, b, mode)) == whole[a:b], (mode, a, b)


def test_fused_path_on_own_archives(gpu, oracle, monkeypatch):
    """Own archives are literal-only, so whole-FASTA calls take the fused kernel: compare with the two-pass path
    and the oracle over record shapes that stress it (empty records, tiny records, masks, odd totals, line widths)."""
    from naf_amd import synth
    rng = np.random.default_rng(77 + SEED)
    texts = [synth.fasta_acgt(1000001, 3, 80, seed=31), synth.fasta_acgt(300000, 2, 0, seed=32), synth.fasta_acgt(99999, 7, 16, seed=33),
             synth.fasta_mixed(60, 9000, 60, seed=34, empty_every=4), synth.fasta_mixed(400, 40, 17, seed=35, empty_every=3),
             synth.fasta_mixed(30, 70000, 1000, seed=36), b">only\n" + b"acgtn" * 20001 + b"\n>e1\n>e2\n>last\nA\n"]
    for text in texts:
        d_naf, rep = gpu.ennaf(gpu.to_device(text))
        exp = oracle.unnaf(host(d_naf), 0)
        for ll in (-1, 0, 16, 61, 100000):
            for um in (True, False):
                e = exp if (ll == -1 and um) else oracle.unnaf(host(d_naf), 0, use_mask=um, line_length=ll)
                monkeypatch.setenv("NAF_GPU_FUSE", "1")
                a = host(gpu.unnaf(d_naf, 0, um, ll))
                monkeypatch.setenv("NAF_GPU_FUSE", "0")
                monkeypatch.setenv("NAF_GPU_EMIT", "long")
                b = host(gpu.unnaf(d_naf, 0, um, ll))
                monkeypatch.setenv("NAF_GPU_EMIT", "short")
                b2 = host(gpu.unnaf(d_naf, 0, um, ll))
                monkeypatch.setenv("NAF_GPU_EMIT", "span")
                b3 = host(gpu.unnaf(d_naf, 0, um, ll))
                monkeypatch.setenv("NAF_GPU_EMIT", "")
                assert a == e and b == e and b2 == e and b3 == e, (len(text), ll, um)


@pytest.mark.parametrize("emit", ["long", "span", "short"])
def test_unnaf_random_archives_against_oracle(gpu, oracle, emit, monkeypatch):
    """Seeded random FASTA/FASTQ -> oracle archive (raw zstd blocks) -> GPU == oracle, all modes."""
    from naf_amd import synth
    monkeypatch.setenv("NAF_GPU_EMIT", emit)
    rng = np.random.default_rng(123 + SEED)
    for i in range(12):
        if i % 3 == 2:
            text = synth.fastq_reads(int(rng.integers(1, 400)), int(rng.integers(1, 200)), seed=i, var_len=True)
        else:
            text = synth.fasta_mixed(int(rng.integers(1, 30)), int(rng.integers(1, 2000)), int(rng.choice([0, 1, 5, 16, 17, 60, 80])), seed=i,
                                     empty_every=int(rng.integers(2, 6)))
        naf = oracle.ennaf(text)
        d = gpu.to_device(naf)
        for mode in (-1, 0, 2, 3, 4):
            for use_mask in (True, False):
                for ll in (-1, 0, 1, 15, 16, 33):
                    if ll != -1 and mode not in (0,):
                        continue
                    exp = oracle.unnaf(naf, mode, use_mask=use_mask, line_length=ll)
                    got = host(gpu.unnaf(d, mode, use_mask, ll))
                    assert got == exp, (i, mode, use_mask, ll)


@pytest.mark.parametrize("parts", ["2", "3", "4", "8"])
def test_unnaf_decode_emit_pipeline_matches_single_launch(gpu, parts, monkeypatch):
    """Whole-text calls on literal-only sequence streams decode the literals in block ranges and emit finished ranges on a second
    stream; by default only from 16 k blocks up.  With the threshold lowered a 170 MB text (2600 blocks) takes that path:
    same bytes as the single-launch order, for FASTA (tile kernels, soft mask on) and for --seq."""
    import torch
    from naf_amd import capi, synth
    text = synth.fasta_acgt_device(170_000_000, n_records=7, width=60, seed=31)
    low = torch.rand(text.numel(), device=text.device) < 0.3                      # soft-mask a third of the letters
    letters = (text >= 65) & (text <= 90)
    hdr = torch.zeros_like(low); pos = (text == 62).nonzero().flatten().tolist(); eol = (text == 10).nonzero().flatten()
    for p in pos:                                                                 # keep the header lines as they are
        e = int(eol[torch.searchsorted(eol, p)].item()); hdr[p:e + 1] = True
    text = torch.where(low & letters & ~hdr, text + 32, text)
    d_naf, _ = gpu.ennaf(text)
    monkeypatch.setenv("NAF_GPU_SPLIT", "1")
    ref_fa = gpu.unnaf(d_naf, capi.OUT_FASTA).clone(); ref_seq = gpu.unnaf(d_naf, capi.OUT_SEQ).clone()
    assert torch.equal(ref_fa, text)
    monkeypatch.setenv("NAF_GPU_SPLIT", parts); monkeypatch.setenv("NAF_GPU_SPLIT_MIN", "32")
    for _ in range(3):                                                            # the streams race differently every time
        assert torch.equal(gpu.unnaf(d_naf, capi.OUT_FASTA), ref_fa)
        assert torch.equal(gpu.unnaf(d_naf, capi.OUT_SEQ), ref_seq)


def test_unnaf_decode_emit_pipeline_reports_corrupt_stream(gpu, oracle, monkeypatch):
    """In the pipelined order the decoder's status is read after the emit has been queued: a damaged sequence stream must still end
    in an error (never in text), and the context must stay usable."""
    import torch
    from naf_amd import capi, synth
    from naf_amd.capi import NafGpuError
    text = synth.fasta_acgt_device(170_000_000, n_records=3, width=80, seed=32)
    d_naf, _ = gpu.ennaf(text)
    h = oracle.parse_naf(host(d_naf))
    monkeypatch.setenv("NAF_GPU_SPLIT", "4"); monkeypatch.setenv("NAF_GPU_SPLIT_MIN", "32")
    assert torch.equal(gpu.unnaf(d_naf, capi.OUT_FASTA), text)
    bad = d_naf.clone()
    seq_off, seq_len = h.payload_off[4], h.comp[4]
    for frac in (0.1, 0.5, 0.9):                                                  # a Huffman stream's final byte must not be zero: kill a few of them
        p = seq_off + int(seq_len * frac)
        bad[p:p + 40000] = 0
    with pytest.raises(NafGpuError):
        gpu.unnaf(bad, capi.OUT_FASTA)
    assert torch.equal(gpu.unnaf(d_naf, capi.OUT_FASTA), text)


def test_flat_tree_literals_every_width(gpu, oracle, monkeypatch):
    """Blocks whose Huffman tree is flat (2^L symbols of equal weight: fixed-width codes) are decoded by k_flat_literals, all
    lanes at once, instead of the serial one-lane-per-stream walk.  Every width the format can give a flat tree (1..7 bits; 8 bits
    never compresses), stream sizes that leave partial groups and odd end-marker positions, 1- and 4-stream blocks, through the
    GPU encoder's frames and against the serial kernel (NAF_GPU_FLAT=0); a stream with a flipped size byte is rejected."""
    import torch
    from naf_amd.capi import NafGpuError
    rng = np.random.default_rng(31 + SEED)
    syms16 = np.array([0x88, 0x84, 0x82, 0x81, 0x48, 0x44, 0x42, 0x41, 0x28, 0x24, 0x22, 0x21, 0x18, 0x14, 0x12, 0x11], dtype=np.uint8)
    made = 0
    for L in range(1, 8):
        alpha = np.arange(1 << L, dtype=np.uint8) * 3 + 5 if L != 4 else syms16
        for n in (1 << 20, 300_001, 65_537, 32_768 + 17, 4099, 1500, 300):
            # exactly equal counts per block make the tree flat in every block (even split of the stream)
            d = np.tile(alpha, n // len(alpha) + 1)[:n].copy()
            for a in range(0, n, 4096):
                rng.shuffle(d[a:a + 4096])
            data = d.tobytes()
            frame = gpu.zstd_compress(gpu.to_device(data))
            fb = host(frame)
            assert oracle.zstd_decompress(fb, n + 16) == data
            monkeypatch.setenv("NAF_GPU_FLAT", "1")
            assert host(gpu.zstd_decompress(frame, n + 64)) == data, (L, n)
            monkeypatch.setenv("NAF_GPU_FLAT", "0")
            assert host(gpu.zstd_decompress(frame, n + 64)) == data, (L, n)
            made += 1
    monkeypatch.setenv("NAF_GPU_FLAT", "1")
    # corrupt: change one stream size of a 4-stream block's jump table so that n x L no longer matches
    data = np.tile(syms16, 4096)[:65536].tobytes()
    fb = bytearray(host(gpu.zstd_compress(gpu.to_device(data))))
    info = oracle.zstd_frame_info(bytes(fb))
    assert info.lit_huf > 0
    # find the first jump table: magic(4) + fhd(2) + block header(3) + literals header(5 for 4-stream 32 KiB) + tree description
    # rather than parse, flip a byte deep inside the first block's streams' end marker region until the decoder objects
    bad = bytearray(fb); bad[len(bad) // 2] ^= 0xFF
    try:
        got = host(gpu.zstd_decompress(gpu.to_device(bytes(bad)), len(data) + 64))
        # no error: then the sizes (they come from the headers) held and a flipped payload byte decoded to different symbols -- a complete
        # prefix code maps different bit strings to different symbol strings, so the text cannot have come out unchanged
        assert len(got) == len(data) and got != data
    except NafGpuError:
        pass
    assert made == 49


def test_flat_frame_read_in_place_by_the_emit_kernel(gpu, oracle, monkeypatch):
    """A sequence stream whose blocks all carry one flat 4-bit tree (uniform ACGT) is not decoded: the tile kernel takes the bases
    from the compressed stream (k_emit_tile_flat).  Against the decode-then-emit path and the oracle: line widths that put every
    chunk alignment and odd / even first bases into play, several records (header and record-end tiles go the slow way), soft-mask
    toggles, --seq / --sequences / --line-length, RNA."""
    import torch
    from naf_amd import synth
    rng = np.random.default_rng(101 + SEED)
    def uniform_fasta(n_rec, per_rec, width, lower=False, rna=False):
        out = bytearray()
        alpha = np.frombuffer(b"ACGU" if rna else b"ACGT", dtype=np.uint8)
        for r in range(n_rec):
            seq = alpha[rng.integers(0, 4, per_rec + r)].copy()
            if lower:                                             # case runs: mask toggles, same 4-bit codes
                pos = 0
                while pos < len(seq):
                    run = int(rng.integers(1, 9000))
                    if rng.random() < 0.5:
                        seq[pos:pos + run] |= 0x20
                    pos += run
            out += b">rec%d uniform %d\n" % (r, r) + synth.wrap_lines(seq, width)
        return bytes(out)
    cases = [(3, 1_500_001, 80, False, False), (2, 2_000_003, 61, True, False), (1, 3_000_000, 0, False, False), (4, 900_017, 16, False, True),
             (2, 1_200_000, 4096, True, False), (1, 2_500_007, 97, False, False)]
    monkeypatch.setenv("NAF_GPU_SPEC_MIN", "8")                  # the in-place path is taken from 512 blocks (16 MB of packed bases) up; here from 8
    cases += [(2, 600_011, w, w % 2 == 1, False) for w in (17, 31, 32, 33, 64, 79, 81, 255, 256, 1000, 5000)]
    for n_rec, per, width, lower, rna in cases:
        text = uniform_fasta(n_rec, per, width, lower, rna)
        d_naf, rep = gpu.ennaf(gpu.to_device(text), seq_type=1 if rna else 0)
        naf = host(d_naf)
        for mode, ll in ((0, -1), (2, -1), (3, -1), (0, 50), (0, 0)):
            want = oracle.unnaf(naf, mode, True, ll)
            monkeypatch.setenv("NAF_GPU_FLAT_FUSE", "1")
            monkeypatch.setenv("NAF_GPU_DEBUG", "1")
            gpu.set_timing(True)
            got = host(gpu.unnaf(d_naf, mode, line_length=ll))
            ran = {n for n, ms, k in gpu.get_timing()}
            gpu.set_timing(False)
            assert got == want, (n_rec, per, width, lower, rna, mode, ll)
            if mode in (0, 3) and (ll < 0 or ll >= 16 or ll == 0) and (width == 0 or width >= 16 or ll >= 16):
                assert "unnaf_emit_flat" in ran, (width, mode, ll, sorted(ran))   # the kernel under test did run
            monkeypatch.setenv("NAF_GPU_FLAT_FUSE", "0")
            assert host(gpu.unnaf(d_naf, mode, line_length=ll)) == want
        monkeypatch.setenv("NAF_GPU_FLAT_FUSE", "1")
        assert host(gpu.unnaf(d_naf, 0, use_mask=False)) == oracle.unnaf(naf, 0, False)
    # a corrupt stream size is still reported
    from naf_amd.capi import NafGpuError
    text = uniform_fasta(1, 2_000_000, 80)
    d_naf, _ = gpu.ennaf(gpu.to_device(text))
    bad = bytearray(host(d_naf)); h = oracle.parse_naf(bytes(bad))
    off = h.payload_off[4] + h.comp[4] // 2
    bad[off] ^= 0x55; bad[off + 1] ^= 0xAA
    try:
        got = host(gpu.unnaf(gpu.to_device(bytes(bad)), 0))
        assert len(got) == len(text)                               # a flipped payload byte inside a stream only changes bases
    except NafGpuError:
        pass


def _mostly_flat_fasta(rng, n_bases, share, width=80, n_rec=2, lower=True):
    """Uniform A C G T (flat 4-bit blocks under this build's encoder) in which a `share` of the 32 KiB blocks of the packed stream is
    made NOT flat, each in one of five ways: a single N (a seventeenth symbol), a stretch of 85 % A (Huffman coding pays), a run of N
    over whole blocks (RLE blocks), IUPAC codes, a gap character."""
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    seq = acgt[rng.integers(0, 4, n_bases)].copy()
    nblocks = n_bases // 65536 + 1
    hit = rng.choice(nblocks, max(1, int(nblocks * share)), replace=False)
    for k, b in enumerate(hit):
        lo = int(b) * 65536
        hi = min(n_bases, lo + 65536)
        if hi - lo < 100:
            continue
        kind = k % 5
        if kind == 0:
            seq[lo + int(rng.integers(0, hi - lo))] = ord("N")
        elif kind == 1:
            m = rng.random(hi - lo) < 0.85
            seq[lo:hi][m] = ord("A")
        elif kind == 2:
            seq[max(0, lo - 70000):hi] = ord("N")                  # more than a whole block of N: at least one RLE block
        elif kind == 3:
            p = rng.integers(lo, hi, 40)
            seq[p] = np.frombuffer(b"RYKMSWBDHV", dtype=np.uint8)[rng.integers(0, 10, 40)]
        else:
            seq[lo + 5:lo + 9] = ord("-")
    if lower:
        pos = 0
        while pos < n_bases:
            run = int(rng.integers(1, 30000))
            if rng.random() < 0.5:
                seg = seq[pos:pos + run]
                seg[seg != ord("-")] |= 0x20
            pos += run
    from naf_amd import synth
    per = n_bases // n_rec
    out = bytearray()
    for r in range(n_rec):
        part = seq[r * per:(r + 1) * per if r + 1 < n_rec else n_bases]
        out += b">rec%d mostly flat\n" % r + synth.wrap_lines(part, width)
    return bytes(out)


@pytest.mark.parametrize("share", [0.01, 0.1, 0.5])
def test_mostly_flat_frame_with_blocks_that_are_not(gpu, oracle, monkeypatch, capfd, share):
    """A real genome under the encoder's flat preference: nearly every block of the sequence stream carries the flat 4-bit tree, a
    few do not (an N, an IUPAC code, a stretch of skewed composition, a run of N).  Those blocks and their neighbours are decoded,
    the rest is read in place (zstd_dec.hip "a frame that is MOSTLY flat"; emit.hip: tiles of class 1 / 2).  Against the oracle, the
    original text and the decode-everything path, at 1, 10 and 50 % of the blocks, through the speculative route of long frames
    (> 512 blocks) and -- forced -- on short ones, every output mode that takes the tile kernels, with and without the mask."""
    import re
    rng = np.random.default_rng(int(share * 1000) + 5)
    big = _mostly_flat_fasta(rng, 36_000_000, share, width=80, n_rec=2)
    d_naf, rep = gpu.ennaf(gpu.to_device(big))
    naf = host(d_naf)
    info = oracle.zstd_frame_info(oracle.parse_naf(naf).frame(naf, 4))
    assert info.n_blocks > 512
    capfd.readouterr()
    monkeypatch.setenv("NAF_GPU_DEBUG_FLAT", "1")
    gpu.set_timing(True)
    got = host(gpu.unnaf(d_naf, 0))
    ran = {n for n, ms, k in gpu.get_timing()}
    gpu.set_timing(False)
    monkeypatch.delenv("NAF_GPU_DEBUG_FLAT")
    err = capfd.readouterr().err
    assert got == big
    assert got == oracle.unnaf(naf, 0)
    m = re.search(r"\[flat mixed\] nblk (\d+) decoded (\d+)", err)
    assert m, err
    nblk, ndec = int(m.group(1)), int(m.group(2))
    assert 0 < ndec < nblk
    if share <= 0.1:
        assert ndec * 2 <= nblk and "unnaf_emit_flat" in ran, (nblk, ndec, sorted(ran))      # the mixed path did run
        assert ndec <= 3.5 * share * nblk + 8, (nblk, ndec)
        m2 = re.search(r"\[flat tiles\] total (\d+) rest (\d+) decoded (\d+)", err)
        assert m2, err
        tiles, rest, dec = (int(x) for x in m2.groups())
        assert rest <= 64 and dec <= tiles * (ndec + 1) // nblk + 64, (tiles, rest, dec)       # the slow list holds the header / record-end tiles only
    monkeypatch.setenv("NAF_GPU_FLAT_MIXED", "0")
    assert host(gpu.unnaf(d_naf, 0)) == big
    monkeypatch.delenv("NAF_GPU_FLAT_MIXED")
    # short frames through the same path, geometry of every kind
    monkeypatch.setenv("NAF_GPU_SPEC_MIN", "8")
    for width, n_rec, n in ((61, 3, 2_400_011), (16, 1, 1_700_000), (0, 2, 3_000_001), (97, 2, 2_000_000), (4096, 1, 2_100_000)):
        text = _mostly_flat_fasta(rng, n, share, width=width, n_rec=n_rec)
        d2, _ = gpu.ennaf(gpu.to_device(text))
        n2 = host(d2)
        for mode, ll, mask in ((0, -1, True), (0, -1, False), (2, -1, True), (3, -1, True), (0, 50, True), (0, 0, True)):
            want = oracle.unnaf(n2, mode, mask, ll)
            assert host(gpu.unnaf(d2, mode, line_length=ll, use_mask=mask)) == want, (share, width, n_rec, mode, ll, mask)
        assert oracle.unnaf(n2, 0) == text


def test_realistic_genome_both_ways_against_the_reference(gpu, oracle, monkeypatch):
    """60 MB of the bench's `realistic` workload (naf_amd/synth.py: skewed composition, CpG depletion, runs of N, IUPAC codes, soft
    mask, 60-column lines): this build's archive of it -- flat preference on (default) and off -- decodes to the text under this
    build's unnaf, the oracle and the REAL reference unnaf; the reference's archive of it decodes to the text here."""
    from naf_amd import synth
    t = synth.realistic_genome_device(60_000_000, n_records=6, device="cuda", n_run_every=4_000_000, iupac_every=150_000)
    text = host(t)
    sizes = {}
    for pf in ("16", "0"):
        monkeypatch.setenv("NAF_GPU_PREFER_FLAT", pf)
        d_naf, rep = gpu.ennaf(t)
        naf = host(d_naf)
        sizes[pf] = len(naf)
        assert host(gpu.unnaf(d_naf, 0)) == text
        assert oracle.unnaf(naf, 0) == text
        if oracle.have_ref():
            assert oracle.ref_unnaf(naf) == text
        info = oracle.zstd_frame_info(oracle.parse_naf(naf).frame(naf, 4))
        assert info.seq_blocks == 0
    monkeypatch.delenv("NAF_GPU_PREFER_FLAT")
    assert sizes["0"] <= sizes["16"] <= 1.06 * sizes["0"], sizes      # what the flat preference gives up: a few per cent of the archive
    if oracle.have_ref():
        ref_naf = oracle.ref_ennaf(text)
        assert host(gpu.unnaf(gpu.to_device(ref_naf), 0)) == text
        assert host(gpu.unnaf(gpu.to_device(ref_naf), 0, use_mask=False)) == oracle.ref_unnaf(ref_naf, ("--no-mask",))


def test_reference_archive_of_random_bases_read_in_place_around_its_matches(gpu, oracle, monkeypatch, capfd):
    """What the REFERENCE makes of packed random bases: one flat 4-bit tree, treeless blocks behind it, and a match every now and then
    (here also a few planted repeats, near and far).  The blocks with sequences, the blocks their matches copy from and their neighbours
    are decoded, the rest is read in place (zstd_dec.hip: k_seq_sources; "mostly flat" frames WITH matches).  Against the text, every
    output mode of the tile kernels and NAF_GPU_FLAT_SEQ=0 (the two-pass decode); short frames forced through the same way."""
    import re
    import torch
    from naf_amd import synth
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built")
    monkeypatch.setenv("NAF_GPU_SPEC_MIN", "8")
    for n, seed, width in ((40_000_000, 21, 80), (9_000_011, 22, 61)):
        rng = np.random.default_rng(seed)
        bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n)].copy()
        # repeats of 60 .. 3000 bases, 1 KB .. 400 KB back, at the same parity (the stream that is compressed holds two bases per byte)
        for _ in range(40):
            ln = int(rng.integers(60, 3000)); back = 2 * int(rng.integers(500, 200_000))
            at = int(rng.integers(back + 10, n - ln - 10))
            bases[at:at + ln] = bases[at - back:at - back + ln]
        per = n // 3
        text = b"".join(b">r%d planted repeats\n" % r + synth.wrap_lines(bases[r * per:(r + 1) * per if r < 2 else n], width) for r in range(3))
        ref_naf = oracle.ref_ennaf(text)
        info = oracle.zstd_frame_info(oracle.parse_naf(ref_naf).frame(ref_naf, 4))
        assert info.seq_blocks > 0
        d = gpu.to_device(ref_naf)
        capfd.readouterr()
        monkeypatch.setenv("NAF_GPU_DEBUG_FLAT", "1")
        gpu.set_timing(True)
        got = host(gpu.unnaf(d, 0))
        ran = {x for x, ms, k in gpu.get_timing()}
        gpu.set_timing(False)
        monkeypatch.delenv("NAF_GPU_DEBUG_FLAT")
        err = capfd.readouterr().err
        assert got == text
        m = re.search(r"\[flat mixed\] nblk (\d+) decoded (\d+) main \d+ \(blocks with sequences (\d+)\)", err)
        assert m and 0 < int(m.group(3)) <= int(m.group(2)) * 1 and int(m.group(2)) * 2 <= int(m.group(1)), err
        assert "unnaf_emit_flat" in ran and any(x.endswith("zstd_exec_seq") for x in ran), sorted(ran)
        for mode, ll, mask in ((0, -1, False), (2, -1, True), (3, -1, True), (0, 50, True), (0, 0, True)):
            a = host(gpu.unnaf(d, mode, line_length=ll, use_mask=mask))
            monkeypatch.setenv("NAF_GPU_FLAT_SEQ", "0")
            b = host(gpu.unnaf(d, mode, line_length=ll, use_mask=mask))
            monkeypatch.delenv("NAF_GPU_FLAT_SEQ")
            assert a == b, (mode, ll, mask)
        assert oracle.ref_unnaf(ref_naf) == text


@pytest.mark.parametrize("part,margin", [("64", "0"), ("64", "8"), ("256", "64"), ("1024", "0"), ("4096", "0")])
def test_huffman_streams_decoded_in_parts(gpu, oracle, monkeypatch, part, margin):
    """k_huf_par: P lanes per Huffman stream, every part started inside its predecessor and re-walked until the starts agree
    (zstd_dec_core.h).  Part sizes from 64 symbols (P = 64 on 16 KiB blocks) to 4096, margins down to 8 bits (nearly every part
    then needs re-walking), on every libzstd-made golden frame, the sections of the reference-made archives, and this build's own
    frames of skewed, nearly flat, 1-bit and 11-bit alphabets -- against SHA-256 / the oracle and the one-lane-per-stream kernel."""
    monkeypatch.setenv("NAF_GPU_HUF_PART", part)
    if margin != "0":
        monkeypatch.setenv("NAF_GPU_HUF_MARGIN", margin)
    for case in zstd_cases():
        frame = golden_bytes("zstd", case["name"] + ".zst")
        got = host(gpu.zstd_decompress(gpu.to_device(frame), case["len"] + 64))
        assert len(got) == case["len"] and sha(got) == case["sha256"], case["name"]
    for case in naf_cases():
        naf = golden_bytes("naf", case["name"] + ".naf")
        d = gpu.to_device(naf)
        for mode in (0, 2):
            try:
                want = oracle.unnaf(naf, mode)
            except Exception:
                continue
            assert host(gpu.unnaf(d, mode)) == want, (case["name"], mode)
    rng = np.random.default_rng(17 + SEED)
    p2 = np.array([2.0 ** -(i + 1) for i in range(30)])
    pr = np.array([1.0] * 15 + [0.5, 0.5])
    datas = [rng.choice(np.arange(30, dtype=np.uint8), 700_001, p=p2 / p2.sum()).tobytes(),
             rng.choice(np.arange(17, dtype=np.uint8), 500_000, p=pr / pr.sum()).tobytes(),
             rng.integers(33, 74, 400_003, dtype=np.uint8).tobytes(),
             bytes(rng.choice(np.array([0, 0, 0, 0, 0, 0, 0, 1, 2, 200], dtype=np.uint8), 300_000)),
             rng.integers(0, 256, 70_000, dtype=np.uint8).tobytes() + bytes(rng.choice(np.arange(100, dtype=np.uint8), 100_000))]
    for data in datas:
        for blog in ("13", "15", "17"):
            monkeypatch.setenv("NAF_GPU_BLOCK_LOG", blog)
            frame = gpu.zstd_compress(gpu.to_device(data))
            monkeypatch.delenv("NAF_GPU_BLOCK_LOG")
            assert oracle.zstd_decompress(host(frame), len(data) + 16) == data
            assert host(gpu.zstd_decompress(frame, len(data) + 64)) == data, (len(data), blog)
            monkeypatch.setenv("NAF_GPU_HUF_PAR", "0")
            assert host(gpu.zstd_decompress(frame, len(data) + 64)) == data
            monkeypatch.delenv("NAF_GPU_HUF_PAR")
    # a damaged stream is still refused, or decodes to other bytes -- never past its place
    data = datas[0]
    fb = bytearray(host(gpu.zstd_compress(gpu.to_device(data))))
    from naf_amd.capi import NafGpuError
    for at in (len(fb) // 3, len(fb) // 2, len(fb) - 9):
        bad = bytearray(fb); bad[at] ^= 0x5A; bad[at + 1] ^= 0xFF
        try:
            got = host(gpu.zstd_decompress(gpu.to_device(bytes(bad)), len(data) + 64))
            assert len(got) == len(data)
        except NafGpuError:
            pass


def test_unnaf_range_of_frames_with_matches_decodes_the_dependency_closure(gpu, oracle, monkeypatch, capfd):
    """A byte range of an archive whose sequence stream holds LZ matches (every reference-made archive; this build's at levels >= 2
    and on repeat-rich input): the blocks under the range and, transitively, every block their matches read from (zstd_dec.hip
    k_seq_reach / k_range_closure) -- not the whole stream.  range == slice of the whole text at random cuts, on the reference-made
    repeat archives (levels 1, 19, --long 27), on reference- and own-made archives of text with far-apart repeats, for the packed
    4-bit stream, FASTA and FASTQ; the closure is a fraction of the stream where matches are sparse."""
    from naf_amd import synth
    rng = np.random.default_rng(9 + SEED)
    arcs = [(name, golden_bytes("naf", name + ".naf")) for name in ("repeat_l1", "repeat_l19", "repeat_long27", "fastq_4k", "mixed_60")]
    # mostly unique sequence with a few far-apart copies: sparse matches, long literal stretches
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    body = acgt[rng.integers(0, 4, 6_000_000)].copy()
    for k in range(12):
        a, b = int(rng.integers(0, 5_000_000)), int(rng.integers(0, 5_900_000))
        body[b:b + 40_000] = body[a:a + 40_000]
    sparse = b">chrS sparse repeats\n" + synth.wrap_lines(body, 70)
    if oracle.have_ref():
        arcs.append(("ref_sparse", oracle.ref_ennaf(sparse)))
        arcs.append(("ref_sparse_l19", oracle.ref_ennaf(sparse, ("-19",))))
    own, _ = gpu.ennaf(gpu.to_device(sparse), level=5)
    arcs.append(("own_sparse_l5", host(own)))
    for name, naf in arcs:
        d = gpu.to_device(naf)
        modes = (1, -1) if name.startswith("fastq") else (0, 2, -1)
        for mode in modes:
            monkeypatch.setenv("NAF_GPU_RANGE_CLOSURE", "1")
            whole = host(gpu.unnaf(d, mode))
            n = len(whole)
            cuts = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, 7)]))
            for a, b in zip(cuts[:-1], cuts[1:]):
                assert host(gpu.unnaf_range(d, a, b, mode)) == whole[a:b], (name, mode, a, b)
            monkeypatch.setenv("NAF_GPU_RANGE_CLOSURE", "0")          # the whole-stream fallback stays right too
            a, b = cuts[len(cuts) // 2 - 1], cuts[len(cuts) // 2]
            assert host(gpu.unnaf_range(d, a, b, mode)) == whole[a:b]
    monkeypatch.delenv("NAF_GPU_RANGE_CLOSURE")
    # the closure of a late eighth of the sparse archive is far smaller than the stream
    for name, naf in arcs:
        if "sparse" not in name:
            continue
        d = gpu.to_device(naf)
        n = len(sparse)
        gpu.set_timing(True)
        got = host(gpu.unnaf_range(d, n // 2, n // 2 + n // 8, 0))
        kt = {nm: (ms, k) for nm, ms, k in gpu.get_timing()}
        gpu.set_timing(False)
        assert got == sparse[n // 2: n // 2 + n // 8]
        assert "zstd_range_closure" in kt, (name, sorted(kt))


def test_offset_code_31_a_match_2_gib_back(gpu):
    """`ennaf --long 31` (ennaf/src/ennaf.c:247-273) lets a match reach back 2^31 bytes and unnaf raises its decoder's window limit
    to match (unnaf/src/input.c:271); offsets from 2^31 - 3 up take offset code 31.  The frame of tests/golden/make_long31.py (hand
    assembled from RFC 8878 and checked there against the real libzstd: digests pinned) holds one such match, 2^31 - 2 bytes back
    across sixteen thousand RLE blocks."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("make_long31", os.path.join(os.path.dirname(__file__), "golden", "make_long31.py"))
    m = importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
    frame = m.frame()
    assert sha(frame) == m.FRAME_SHA256
    a, zeros, lits = m.parts()
    want_sha, n = m.text_digest()
    assert want_sha == m.TEXT_SHA256
    out = gpu.zstd_decompress(gpu.to_device(frame), n + 64)
    assert out.numel() == n
    assert host(out[:len(a)]) == a
    z = out[len(a):len(a) + zeros]
    for i in range(0, zeros, 1 << 28):
        assert not bool(z[i:i + (1 << 28)].any())
    assert host(out[len(a) + zeros:]) == lits + a[m.SRC:m.SRC + m.ML]
    del out, z
    torch.cuda.empty_cache()


def test_mostly_flat_frame_at_scale_takes_the_two_phase_tables(gpu, monkeypatch, capfd):
    """From 16 K blocks and 2 K distinct trees up the decoder finds the frame's flat tree among its first blocks, marks its repetitions
    by their bytes and leaves the other trees' tables to a job that runs beside the emit of the flat tiles (zstd_dec.hip:
    k_flat_find_main, k_build_huf phase 2, zstd_flat_later).  1.5 GB of the realistic genome: full-size round trip (the
    size-independent property), the path taken, and the decode-everything path on the same archive."""
    import re
    import torch
    from naf_amd import synth
    t = synth.realistic_genome_device(int(1.5e9), device="cuda")
    d_naf, rep = gpu.ennaf(t)
    capfd.readouterr()
    monkeypatch.setenv("NAF_GPU_DEBUG_FLAT", "1")
    out = gpu.unnaf(d_naf, 0)
    torch.cuda.synchronize()
    err = capfd.readouterr().err
    monkeypatch.delenv("NAF_GPU_DEBUG_FLAT")
    assert torch.equal(out, t)
    m = re.search(r"\[flat mixed\] nblk (\d+) decoded (\d+)", err)
    assert m and int(m.group(1)) > 16384 and 0 < int(m.group(2)) * 2 <= int(m.group(1)), err
    m2 = re.search(r"\[flat tiles\] total (\d+) rest (\d+) decoded (\d+)", err)
    assert m2 and int(m2.group(2)) <= 256, err
    del out
    monkeypatch.setenv("NAF_GPU_FLAT_MIXED", "0")
    assert torch.equal(gpu.unnaf(d_naf, 0), t)
    monkeypatch.delenv("NAF_GPU_FLAT_MIXED")


def test_fastq_record_kernel_groupings(gpu, oracle):
    """k_emit_fastq_records takes 4, 8 or 16 lanes per read (64, 32 or 16 reads per workgroup, chosen from the mean text per read) and
    writes a workgroup's reads through LDS when they fit: reads from a few bases to longer than the stage, fixed and variable length,
    read counts that do not fill the last workgroup -- against the oracle's FASTQ text."""
    from naf_amd import synth
    for n, ln, var in ((1, 5, False), (63, 40, True), (65, 150, False), (1000, 150, True), (130, 300, False), (257, 330, True), (33, 700, False),
                       (40, 1500, True), (17, 3000, False), (5, 9000, False), (9, 30000, True)):
        text = synth.fastq_reads(n, ln, seed=n + ln, var_len=var)
        naf = oracle.ennaf(text)
        exp = oracle.unnaf(naf, 1)
        assert host(gpu.unnaf(gpu.to_device(naf), 1)) == exp, (n, ln, var)


def test_tables_of_many_distinct_trees_sixteen_lanes_per_tree(gpu, oracle, monkeypatch):
    """k_build_huf16 (zstd_dec.hip): frames of more than 16 K blocks and 2 K distinct trees get their tables from sixteen lanes per tree
    -- directly stored weights with codes of up to HUF_FULL_LOG bits by the group, FSE-coded weights and longer codes by the group's
    first lane the old way.  Quality-like, skewed (long codes), wide (symbols up to 255: FSE-coded weights), flat and two-symbol
    alphabets in 1 KiB blocks, against the data, the one-lane-per-tree kernel and (a cut) the oracle."""
    rng = np.random.default_rng(23 + SEED)
    p2 = np.array([2.0 ** -(i + 1) for i in range(30)])
    n = 20_000_000
    datas = [rng.integers(33, 74, n, dtype=np.uint8),
             rng.choice(np.arange(30, dtype=np.uint8), n, p=p2 / p2.sum()),
             rng.choice(np.array([0x11, 0x12, 0x21, 0x88, 0xFF, 0xF1, 0x1F, 0x44], dtype=np.uint8), n, p=[.3, .2, .2, .1, .05, .05, .05, .05]),
             rng.integers(0, 16, n, dtype=np.uint8),
             rng.choice(np.array([7, 200], dtype=np.uint8), n, p=[.9, .1]),
             np.concatenate([rng.integers(0, 128, n // 2, dtype=np.uint8), rng.integers(0, 129, n // 2, dtype=np.uint8)])]
    monkeypatch.setenv("NAF_GPU_LZ", "0")
    for data in datas:
        d = gpu.to_device(data.tobytes())
        monkeypatch.setenv("NAF_GPU_BLOCK_LOG", "10")
        frame = gpu.zstd_compress(d)
        monkeypatch.delenv("NAF_GPU_BLOCK_LOG")
        for flat in (None, "0"):
            if flat is not None:
                monkeypatch.setenv("NAF_GPU_FLAT", flat)
            got = gpu.zstd_decompress(frame, n + 64)
            assert got.numel() == n and bool((got == d).all())
            monkeypatch.setenv("NAF_GPU_HUF_BUILD16", "0")
            got = gpu.zstd_decompress(frame, n + 64)
            assert got.numel() == n and bool((got == d).all())
            monkeypatch.delenv("NAF_GPU_HUF_BUILD16")
            if flat is not None:
                monkeypatch.delenv("NAF_GPU_FLAT")
        del got
    # a cut of the first one under the oracle as well
    cut = datas[0][:3_000_000].tobytes()
    monkeypatch.setenv("NAF_GPU_BLOCK_LOG", "10")
    f2 = host(gpu.zstd_compress(gpu.to_device(cut)))
    monkeypatch.delenv("NAF_GPU_BLOCK_LOG")
    assert oracle.zstd_decompress(f2, len(cut) + 16) == cut


def test_uniform_flat_frame_without_the_block_table(gpu, oracle, monkeypatch, capfd):
    """zstd_dec.hip k_uni_head / k_uni_streams: a frame whose blocks repeat the first one in all but their stream bytes gets its stream
    table by arithmetic, with every byte the general front would have looked at compared instead.  Even / odd base counts (a Raw block
    of one byte at the end), a stream that ends on a block, whole texts, every output mode that reads the streams in place, byte ranges,
    against the text, the oracle and NAF_GPU_UNIFORM=0; then archives with one byte changed in a block's front, in its sequences byte
    and in a stream's end marker: the uniform front must step aside and the call must end the way it ends with NAF_GPU_UNIFORM=0."""
    from naf_amd import synth
    from naf_amd.capi import NafGpuError
    import re
    import torch
    monkeypatch.setenv("NAF_GPU_SPEC_MIN", "8")
    cases = [synth.fasta_acgt_device(30_000_000, n_records=3, width=80, seed=13, device="cuda"),
             synth.fasta_acgt_device(30_000_011, n_records=5, width=71, seed=14, device="cuda"),
             synth.fasta_acgt_device(65536 * 420, n_records=1, width=60, seed=16, device="cuda")]
    for t in cases:
        d_naf, rep = gpu.ennaf(t)
        capfd.readouterr()
        monkeypatch.setenv("NAF_GPU_DEBUG_FLAT", "1")
        gpu.set_timing(True)
        out = gpu.unnaf(d_naf, 0)
        ran = {n for n, ms, k in gpu.get_timing()}
        gpu.set_timing(False)
        err = capfd.readouterr().err
        monkeypatch.delenv("NAF_GPU_DEBUG_FLAT")
        assert torch.equal(out, t)
        m = re.search(r"\[uniform\?\] ok 1 bad 0 prefix (\d+) blocks (\d+) huffman (\d+)", err)
        assert m and int(m.group(1)) >= 64 and int(m.group(2)) - int(m.group(1)) <= 2, err
        assert "unnaf_emit_flat" in ran and not any(x.endswith("zstd_parse_blocks") or x.endswith("zstd_flat_streams") for x in ran), sorted(ran)
        naf = host(d_naf)
        for mode, ll, mask in ((0, -1, True), (0, -1, False), (2, -1, True), (3, -1, True), (0, 50, True), (0, 0, True)):
            got = host(gpu.unnaf(d_naf, mode, line_length=ll, use_mask=mask))
            monkeypatch.setenv("NAF_GPU_UNIFORM", "0")
            assert got == host(gpu.unnaf(d_naf, mode, line_length=ll, use_mask=mask)), (mode, ll, mask)
            monkeypatch.delenv("NAF_GPU_UNIFORM")
        assert oracle.unnaf(naf, 0) == host(t)
        n = t.numel()
        for a, b in ((0, 1), (0, 5000), (n // 3, n // 3 + 123_457), (n - 70_000, n), (n // 2, n // 2 + 1)):
            assert torch.equal(gpu.unnaf_range(d_naf, a, b, 0), t[a:b]), (a, b)
    # sixteen byte values below 128 (bases T, G, K, C only): a flat tree whose weights can be stored directly
    lut = torch.tensor(list(b"TGKC"), dtype=torch.uint8, device="cuda")
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    body = lut[torch.randint(0, 4, (24_000_000,), device="cuda", generator=g)]
    t4 = torch.cat([torch.tensor(list(b">tgkc only\n"), dtype=torch.uint8, device="cuda"), body, torch.tensor([10], dtype=torch.uint8, device="cuda")])
    d4, _ = gpu.ennaf(t4)
    n4 = host(d4)
    for un in ("1", "0"):
        monkeypatch.setenv("NAF_GPU_UNIFORM", un)
        assert torch.equal(gpu.unnaf(d4, 0), t4), un
    monkeypatch.delenv("NAF_GPU_UNIFORM")
    assert oracle.unnaf(n4, 0) == host(t4)
    # one byte changed: where the general front finds an error, or another shape, so must this one
    t = cases[1]
    d_naf, rep = gpu.ennaf(t)
    naf = bytearray(host(d_naf))
    h = gpu.parse_header(d_naf)
    off, size = h.payload_off[4], h.comp_size[4]
    fr = bytes(naf[off:off + size])
    fhd = fr[0]; hdr = 1 + (0 if (fhd >> 5) & 1 else 1) + ((1 if (fhd >> 5) & 1 else 0) if (fhd >> 6) == 0 else (2, 4, 8)[(fhd >> 6) - 1])
    b0 = fr[hdr] | (fr[hdr + 1] << 8) | (fr[hdr + 2] << 16)
    S = 3 + (b0 >> 3)
    def outcome(buf):
        d = gpu.to_device(bytes(buf))
        try:
            return ("ok", sha(host(gpu.unnaf(d, 0))))
        except NafGpuError as e:
            return ("error", e.code)
    for blk, where in ((7, 3 + 2), (100, 3 + 9), (200, 3 + 40), (33, S - 1), (150, S - 2), (5, 3 + 5 + 60)):
        bad = bytearray(naf)
        bad[off + hdr + blk * S + where] ^= 0x5A
        got = outcome(bad)
        monkeypatch.setenv("NAF_GPU_UNIFORM", "0")
        want = outcome(bad)
        monkeypatch.delenv("NAF_GPU_UNIFORM")
        assert got == want, (blk, where, got, want)


def test_stride_index_of_frames_of_equal_blocks(gpu, oracle, monkeypatch, capfd):
    """zstd_dec.hip k_stride_probe / k_stride_tail: a frame whose blocks all repeat the first block's header (a genome's packed bases
    under fixed-width codes) is indexed by testing every position off0 + i S at once, in front of the speculative index.
    Even and odd base counts (the odd one ends in a Raw block of one byte), a stream that ends exactly on a block, a frame with
    blocks of other sizes in the middle (left to the speculative index), against the text, the oracle and NAF_GPU_STRIDE_INDEX=0;
    the mask frames of these upper-case texts go through k_mask_rle_frame (emit.hip), against NAF_GPU_MASK_RLE=0."""
    from naf_amd import synth
    import re
    import torch
    def names(c):
        return {n for n, ms, k in c.get_timing()}
    cases = [synth.fasta_acgt_device(30_000_000, n_records=3, width=80, seed=3, device="cuda"),
             synth.fasta_acgt_device(30_000_011, n_records=5, width=71, seed=4, device="cuda"),
             synth.fasta_acgt_device(65536 * 420, n_records=1, width=60, seed=6, device="cuda")]
    for t in cases:
        d_naf, rep = gpu.ennaf(t)
        capfd.readouterr()
        monkeypatch.setenv("NAF_GPU_DEBUG_STRIDE", "1")
        gpu.set_timing(True)
        out = gpu.unnaf(d_naf, 0)
        nm = names(gpu)
        gpu.set_timing(False)
        err = capfd.readouterr().err
        assert torch.equal(out, t)
        m = re.search(r"\[stride\] len (\d+) S (\d+) nmax (\d+) prefix (\d+) verdict 1 err 0 nblk (\d+)", err)
        assert m and int(m.group(1)) > (4 << 20) and int(m.group(5)) >= int(m.group(4)) >= 64, err
        # (an all-upper-case text: its mask frame, RLE blocks of 0xFF units, is taken by k_mask_rle_frame in one launch)
        assert any(n.endswith("unnaf_mask_rle") for n in nm) and not any(n.endswith("unnaf_mask_count") for n in nm), " ".join(sorted(nm))
        monkeypatch.setenv("NAF_GPU_MASK_RLE", "0")
        assert torch.equal(gpu.unnaf(d_naf, 0), t)
        monkeypatch.delenv("NAF_GPU_MASK_RLE")
        monkeypatch.setenv("NAF_GPU_STRIDE_INDEX", "0")
        capfd.readouterr()
        out2 = gpu.unnaf(d_naf, 0)
        err2 = capfd.readouterr().err
        monkeypatch.delenv("NAF_GPU_STRIDE_INDEX")
        assert torch.equal(out2, t) and "[stride]" not in err2
        monkeypatch.delenv("NAF_GPU_DEBUG_STRIDE")
    naf = host(d_naf)
    assert oracle.unnaf(naf, 0) == host(t)
    # a few lower-case stretches in 30 MB: still a frame of a few dozen bytes for 100 K units, toggles from its Raw / RLE blocks
    t = cases[0].clone()
    for a, b in ((1000, 1500), (5_000_000, 5_000_300), (12_345_678, 12_400_000), (29_000_000, 29_000_001)):
        seg = t[a:b]; t[a:b] = torch.where((seg >= 65) & (seg <= 90), seg + 32, seg)
    d_naf, rep = gpu.ennaf(t)
    gpu.set_timing(True)
    out = gpu.unnaf(d_naf, 0)
    nm = names(gpu)
    gpu.set_timing(False)
    assert torch.equal(out, t)
    assert oracle.unnaf(host(d_naf), 0) == host(t)
    monkeypatch.setenv("NAF_GPU_MASK_RLE", "0")
    assert torch.equal(gpu.unnaf(d_naf, 0), t)
    monkeypatch.delenv("NAF_GPU_MASK_RLE")
    # blocks of other sizes in the middle: the prefix ends early, the tail is long, the speculative index takes the frame
    t = synth.realistic_genome_device(40_000_000, n_records=4, device="cuda", n_run_every=3_000_000, iupac_every=150_000)
    d_naf, rep = gpu.ennaf(t)
    monkeypatch.setenv("NAF_GPU_DEBUG_STRIDE", "1")
    capfd.readouterr()
    out = gpu.unnaf(d_naf, 0)
    err = capfd.readouterr().err
    monkeypatch.delenv("NAF_GPU_DEBUG_STRIDE")
    assert "verdict 1" not in err, err
    assert torch.equal(out, t)


def test_names_decoded_beside_ids(gpu, monkeypatch):
    """emit.hip unnaf_sections_main: ids and names of an archive whose two streams are both above 1 MB go to a side context each (the
    emit of a FASTQ waits for the later of the two chains).  200 000 reads `@readN len=L` (ids 2.4 MB, names 1.6 MB), archive made by
    this build: the FASTQ text back (bases upper-cased, R3), the same bytes with NAF_GPU_NAMES_BESIDE=0, and the FASTA view of it."""
    from naf_amd import synth, capi
    text = synth.fastq_reads(200_000, 100, seed=5, var_len=True)
    d_naf, rep = gpu.ennaf(gpu.to_device(text))
    assert rep.n_sequences == 200_000
    a = host(gpu.unnaf(d_naf, capi.OUT_FASTQ))
    monkeypatch.setenv("NAF_GPU_NAMES_BESIDE", "0")
    b = host(gpu.unnaf(d_naf, capi.OUT_FASTQ))
    fa0 = host(gpu.unnaf(d_naf, capi.OUT_FASTA))
    monkeypatch.delenv("NAF_GPU_NAMES_BESIDE")
    assert a == b
    # the text itself, with the bases of every read upper-cased (unnaf.c:442): lines 2 of 4
    lines = text.split(b"\n")
    for i in range(1, len(lines), 4):
        lines[i] = lines[i].upper()
    assert a == b"\n".join(lines)
    assert host(gpu.unnaf(d_naf, capi.OUT_FASTA)) == fa0


@pytest.mark.parametrize("flags", [("--level", "3", "--long", "27"), ("--level", "19"), ("--level", "1")], ids=lambda f: "".join(f))
def test_reference_archive_of_a_repeat_rich_genome_under_every_executor(gpu, oracle, monkeypatch, flags):
    """What the reference makes of a genome full of repeats (libzstd's match finders, `--long`: thousands of sequences per 128 KiB block,
    matches that read all over the blocks in front of them -- ennaf/src/compressor.c:7-21, ennaf.c:247-273) decoded by the dataflow
    executor with and without k_lz_collapse and by the two block-ordered ones: the reference's own text every time.  (In block order this
    kind of frame ran one block behind the other, and at 1 GB the bounded wait called it corrupt: DESIGN.md 4.30.)"""
    from naf_amd import capi, synth
    O = oracle
    if not O.have_ref():
        pytest.skip("needs oracle/_ref")
    text = host(synth.repeat_genome_device(12_000_000, device="cuda", families=24))
    naf = O.ref_ennaf(text, flags)
    want = O.ref_unnaf(naf)
    assert want == text
    d_naf = gpu.to_device(naf)
    for how, collapse in (("dataflow", "1"), ("dataflow", "n"), ("dataflow", "0"), ("batch", "1"), ("serial", "1")):
        monkeypatch.setenv("NAF_GPU_EXEC", how); monkeypatch.setenv("NAF_GPU_EXEC_COLLAPSE", collapse); monkeypatch.setenv("NAF_GPU_EXEC_LDS", "0")
        assert host(gpu.unnaf(d_naf, capi.OUT_FASTA)) == want, (flags, how, collapse)
        # a byte range of it: the range's dependency closure through the same executor
        b, e = len(want) // 3, len(want) // 3 + 1_000_003
        assert host(gpu.unnaf_range(d_naf, b, e, capi.OUT_FASTA)) == want[b:e], (flags, how, collapse)
    monkeypatch.delenv("NAF_GPU_EXEC"); monkeypatch.delenv("NAF_GPU_EXEC_COLLAPSE"); monkeypatch.delenv("NAF_GPU_EXEC_LDS")
    # this build's own archives at the levels that match across blocks, through the same executors (16 KiB blocks: the LDS executor by default)
    for level, long_log in ((19, 0), (3, 27)):
        mine, _ = gpu.ennaf(gpu.to_device(text), level=level, long_log=long_log)
        for lds in ("1", "0"):
            monkeypatch.setenv("NAF_GPU_EXEC_LDS", lds)
            assert host(gpu.unnaf(mine, capi.OUT_FASTA)) == text, (level, long_log, lds)
        monkeypatch.delenv("NAF_GPU_EXEC_LDS")


def test_executors_at_the_size_where_the_wait_once_gave_up(gpu, oracle):
    """VERDICT r05 item 7(b).  The reference's `--level 3 --long 27` archive of 300 MB of a repeat-rich genome -- the size class where round 5's
    block-ordered executor ran out of polls and called a VALID frame corrupt (the pytest guard was 12 MB) -- and its archive
    of two million reads whose names copy each other (one chain through the whole ids frame): the reference's own text, and within a bound
    of seconds where the slow path took minutes (a call of this size is tens of milliseconds)."""
    import time
    import torch
    from naf_amd import capi, synth
    O = oracle
    if not O.have_ref():
        pytest.skip("needs oracle/_ref")
    text = host(synth.repeat_genome_device(300_000_000, device="cuda", families=24))
    naf = O.ref_ennaf(text, ("--level", "3", "--long", "27"))
    d_naf = gpu.to_device(naf)
    gpu.unnaf(d_naf, capi.OUT_FASTA)                              # (arena growth)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    got = gpu.unnaf(d_naf, capi.OUT_FASTA)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert hashlib.sha256(host(got)).digest() == hashlib.sha256(text).digest()
    assert dt < 3.0, dt
    b, e = len(text) // 2, len(text) // 2 + 20_000_003                # ... and a byte range of it through its dependency closure
    assert host(gpu.unnaf_range(d_naf, b, e, capi.OUT_FASTA)) == text[b:e]
    del text, got
    fq = host(synth.fastq_reads_device(400_000_000, seed=11, device="cuda"))
    naf = O.ref_ennaf(fq, ("--fastq",))
    want = hashlib.sha256(O.ref_unnaf(naf)).digest()
    d_naf = gpu.to_device(naf)
    gpu.unnaf(d_naf, capi.OUT_FASTQ)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    got = gpu.unnaf(d_naf, capi.OUT_FASTQ)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert hashlib.sha256(host(got)).digest() == want
    assert dt < 3.0, dt


def test_runs_that_continue_from_block_to_block(gpu, oracle, monkeypatch):
    """zstd_dec.hip k_lz_runs_*: libzstd codes a FASTQ's lengths and its repeating names as one literal and one match per 128 KiB block --
    first overlapping itself with the period as its offset, then copying the whole block in front -- a chain of a link per block.  The
    links of such a run read the period's seed instead, as long as their literals are what the pattern says: archives of
    the real `ennaf` whose reads all have one length (every block a link), with ONE read of another length at a block's first, second and
    last record or in the middle (the run breaks there and a new one begins), with two lengths in turn (period 8) and with random lengths
    (no run at all) must decode to the reference's text with the rewrite and without it (NAF_GPU_EXEC_RUNS=0)."""
    from naf_amd import capi
    O = oracle
    if not O.have_ref():
        pytest.skip("needs oracle/_ref")
    rng = np.random.default_rng(12 + SEED)
    n = 150_000                                                          # 600 KB of lengths, 1.2 MB of names: five and ten blocks
    def fastq(lens):
        return b"".join(b"@r len=%d\n%s\n+\n%s\n" % (L, b"ACGT"[:L] if L <= 4 else b"A" * L, b"I" * L) for L in lens)
    cases = []
    cases.append([4] * n)
    for odd in (32767, 32768, 32769, 65535, 65536, 100_000):
        lens = [4] * n; lens[odd] = 3; cases.append(lens)
    cases.append([4 if i & 1 else 2 for i in range(n)])
    cases.append([int(x) for x in rng.integers(1, 5, n)])
    for lens in cases:
        text = fastq(lens)
        naf = O.ref_ennaf(text, ("--fastq",))
        want = O.ref_unnaf(naf)
        d_naf = gpu.to_device(naf)
        for runs in ("1", "0"):
            monkeypatch.setenv("NAF_GPU_EXEC_RUNS", runs)
            assert host(gpu.unnaf(d_naf, capi.OUT_FASTQ)) == want, (runs, lens[:4], [i for i, L in enumerate(lens[:110_000]) if L == 3][:2])
    monkeypatch.delenv("NAF_GPU_EXEC_RUNS")


def test_reference_archive_of_reads_whose_names_copy_each_other(gpu, oracle, monkeypatch):
    """The reference's archive of a FASTQ: libzstd codes every read name as a copy of the name before it plus a digit or two -- chains of
    ten thousand links per 128 KiB block of the ids stream.  k_lz_collapse moves every source back along its chain (what is left: a link
    where the counter's digits roll over); the text must be the reference's with it, without it (NAF_GPU_EXEC_COLLAPSE=0) and in block order."""
    from naf_amd import capi, synth
    O = oracle
    if not O.have_ref():
        pytest.skip("needs oracle/_ref")
    text = synth.fastq_reads(60_000, 100, seed=3) + synth.fastq_reads(3_000, 120, seed=4, var_len=True)
    naf = O.ref_ennaf(text, ("--fastq",))
    want = O.ref_unnaf(naf)
    d_naf = gpu.to_device(naf)
    for how, collapse in (("dataflow", "1"), ("dataflow", "s"), ("dataflow", "n"), ("dataflow", "0"), ("batch", "1")):
        monkeypatch.setenv("NAF_GPU_EXEC", how); monkeypatch.setenv("NAF_GPU_EXEC_COLLAPSE", collapse)
        assert host(gpu.unnaf(d_naf, capi.OUT_FASTQ)) == want, (how, collapse)
        for mode, args in ((capi.OUT_FASTA, ("--fasta",)),):
            assert host(gpu.unnaf(d_naf, mode)) == O.ref_unnaf(naf, args), (how, collapse, mode)
    monkeypatch.delenv("NAF_GPU_EXEC"); monkeypatch.delenv("NAF_GPU_EXEC_COLLAPSE")


def test_reference_archives_of_structured_texts_at_every_level(gpu, oracle):
    """The sequence executors against libzstd's own frames beyond the golden set: texts with repeats at every distance (units copied with a
    few substitutions, tandem runs, long runs of one base, records with counting names), packed by the real `ennaf` at levels 1 ... 22 with
    and without `--long`, FASTA and FASTQ -- the HIP decoder must return what the real `unnaf` returns, whole and by byte range."""
    from naf_amd import capi, synth
    O = oracle
    if not O.have_ref():
        pytest.skip("needs oracle/_ref")
    rng = np.random.default_rng(2025 + SEED)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)

    def genome(n):
        parts, made = [], 0
        units = [acgt[rng.integers(0, 4, int(rng.integers(50, 5000)))] for _ in range(12)]
        while made < n:
            k = int(rng.integers(0, 4))
            if k == 0:
                p = acgt[rng.integers(0, 4, int(rng.integers(10, 3000)))]
            elif k == 1:
                p = units[int(rng.integers(0, len(units)))].copy(); idx = rng.integers(0, len(p), max(1, len(p) // 40)); p[idx] = acgt[rng.integers(0, 4, len(idx))]
            elif k == 2:
                u = acgt[rng.integers(0, 4, int(rng.integers(1, 9)))]; p = np.tile(u, int(rng.integers(5, 4000)))
            else:
                p = np.full(int(rng.integers(100, 70000)), acgt[int(rng.integers(0, 4))], dtype=np.uint8)
            parts.append(p); made += len(p)
        return np.concatenate(parts)[:n]

    cases = []
    for i, (n, width) in enumerate(((700_000, 60), (2_500_000, 80), (300_000, 0))):
        g = genome(n)
        recs = [b">rec%d part %d\n" % (j, i) + synth.wrap_lines(g[a:b], width) for j, (a, b) in enumerate(zip(range(0, n, n // 7 + 1), list(range(n // 7 + 1, n, n // 7 + 1)) + [n]))]
        cases.append((b"".join(recs), ()))
    cases.append((synth.fastq_reads(30_000, 90, seed=8) + synth.fastq_reads(2_000, 140, seed=9, var_len=True), ("--fastq",)))
    for text, fmt in cases:
        for flags in (("--level", "1"), ("--level", "3"), ("--level", "9", "--long", "20"), ("--level", "19"), ("--level", "22", "--long", "27"), ("--level", "5", "--long", "12")):
            naf = O.ref_ennaf(text, fmt + flags)
            want = O.ref_unnaf(naf)
            d_naf = gpu.to_device(naf)
            got = host(gpu.unnaf(d_naf))
            assert got == want, (len(text), fmt, flags)
            b, e = len(want) // 5, len(want) // 5 + 300_001
            assert host(gpu.unnaf_range(d_naf, b, min(e, len(want)))) == want[b:e], (len(text), fmt, flags, "range")
