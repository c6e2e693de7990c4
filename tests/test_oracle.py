"""CPU tests: pin the oracle (oracle/*.c) against the reference's own fixtures and against golden
vectors produced by the real reference binaries / the image's libzstd (tests/golden/make_golden.py).
"""
import hashlib

import numpy as np
import pytest

from conftest import golden_bytes, naf_cases, ref_cases, zstd_cases


def sha(b):
    return hashlib.sha256(b).hexdigest()


# ---- known-answer vectors that exist in the reference itself ---------------------
def test_vle_table_from_format_spec(oracle):
    import ctypes as C
    table = {0: "00", 127: "7f", 128: "8100", 129: "8101", 34359738367: "ffffffff7f", 34359738368: "818080808000"}
    for v, hx in table.items():
        buf = C.create_string_buffer(10)
        n = oracle.lib().nafo_vle_write(v, buf)
        assert buf.raw[:n].hex() == hx
        out = C.c_uint64()
        assert oracle.lib().nafo_vle_read(bytes.fromhex(hx), n, C.byref(out)) == n and out.value == v
    out = C.c_uint64()
    assert oracle.lib().nafo_vle_read(b"\x80\x01", 2, C.byref(out)) == -1          # unnaf utils.c:128
    assert oracle.lib().nafo_vle_read(b"\xff" * 10 + b"\x01", 11, C.byref(out)) == -2   # overflow check
    assert oracle.lib().nafo_vle_read(b"\x81", 1, C.byref(out)) == 0               # truncated


def test_nucleotide_code_table(oracle):
    # NAFv2.pdf p.5 == ennaf tables.c:189-197 == inverse of unnaf.c:13
    tab = b"-TGKCYSBAWRDMHVN"
    for code, ch in enumerate(tab):
        assert oracle.pack_4bit(bytes([ch])) == bytes([code])
        assert oracle.pack_4bit(bytes([ch | 0x20]) if ch != ord("-") else b"-") == bytes([code])
        assert oracle.unpack_4bit(bytes([code]), 1) == bytes([ch])
    assert oracle.pack_4bit(b"U") == b"\x01" and oracle.unpack_4bit(b"\x01", 1, rna=True) == b"U"
    assert oracle.pack_4bit(b"AC") == bytes([8 | (4 << 4)])                          # first base low nibble
    for other in b"EFIJLOPQXZ@[0 \x00\xff":
        assert oracle.pack_4bit(bytes([other])) == b"\x0f"


def test_79_byte_example_from_survey(oracle):
    # the archive of tests/small/1.fa as written by the reference
    naf = bytes.fromhex(
        "01f9ec013e200a02" "04090048210000310032" "00" "060b00483100000073657132" "00"
        "080d0048410000" "0a000000" "07000000" "090e0048490000" "0004040101010101" "04"
        "110e0048490000" "48214812ff08f1c005")
    text = golden_bytes("ref_tests", "small", "1.fa")
    assert oracle.unnaf(naf) == golden_bytes("ref_tests", "small", "1-default.out-ref")
    sp = oracle.split_text(text)
    assert sp.ids == b"1\x002\x00" and sp.comments == b"\x00seq2\x00"
    assert sp.lengths == bytes.fromhex("0a00000007000000")
    assert sp.mask == bytes.fromhex("0004040101010101" "04")
    assert sp.seq == bytes.fromhex("48214812ff08f1c005")


# ---- the reference's own test suite, through the oracle ---------------------------------------------
def _seq_type(args, O):
    return O.RNA if "--rna" in args else O.PROTEIN if "--protein" in args else O.TEXT if "--text" in args else O.DNA


def unexpected_report(sp, seq_type_name):
    """stderr text of ennaf's report (process.c:75-96)."""
    out = []
    for key, nm in (("id", "id"), ("comment", "comment"), ("seq", seq_type_name), ("qual", "quality")):
        n = sp.unexpected[key]
        tot = sum(n)
        if not tot:
            continue
        out.append("input has %d unexpected %s characters:\n" % (tot, nm))
        for i in range(32):
            if n[i]:
                out.append("    '\\x%02X': %d\n" % (i, n[i]))
        for i in range(32, 127):
            if n[i]:
                out.append("    '%c': %d\n" % (i, n[i]))
        for i in range(127, 256):
            if n[i]:
                out.append("    '\\x%02X': %d\n" % (i, n[i]))
        if n[256]:
            out.append("    EOF: %d\n" % n[256])
    return "".join(out).encode("latin1")


@pytest.mark.parametrize("case", ref_cases(), ids=lambda c: c["set"] + "/" + c["name"])
def test_reference_suite_through_oracle(oracle, case):
    O = oracle
    text = golden_bytes("ref_tests", case["set"], case["input"])
    ea, ua = case["ennaf_args"], case["unnaf_args"]
    st = _seq_type(ea, O)
    sp = O.split_text(text, st, "--no-mask" in ea)
    naf = O.ennaf(text, st, "--no-mask" in ea)
    if "--charcount" in ua:                                   # output.c:515-605: byte counts of the --seq text
        seq = O.unnaf(naf, O.MODE_SEQ, use_mask="--no-mask" not in ua)
        cnt = [0] * 256
        for b in seq:
            cnt[b] += 1
        lines = ["\\x%02X\t%d\n" % (i, cnt[i]) for i in range(33) if cnt[i]]
        lines += ["%c\t%d\n" % (i, cnt[i]) for i in range(33, 127) if cnt[i]]
        lines += ["\\x%02X\t%d\n" % (i, cnt[i]) for i in range(127, 256) if cnt[i]]
        assert "".join(lines).encode("latin1") == golden_bytes("ref_tests", case["set"], case["name"] + ".out-ref")
        return
    mode = O.MODE_SEQ if "--seq" in ua else O.MODE_SEQUENCES if "--sequences" in ua else -1
    out = O.unnaf(naf, mode, use_mask="--no-mask" not in ua)
    pre = [case["set"], case["name"]]
    assert out == golden_bytes("ref_tests", pre[0], pre[1] + ".out-ref")
    names = {O.DNA: "DNA", O.RNA: "RNA", O.PROTEIN: "protein", O.TEXT: "text"}
    assert unexpected_report(sp, names[st]) == golden_bytes("ref_tests", pre[0], pre[1] + ".e.err-ref")
    assert golden_bytes("ref_tests", pre[0], pre[1] + ".u.err-ref") == b""


# ---- golden archives made by the real reference ------------------------------------------------------
MODES = {"fasta": (0, True, -1), "seq": (2, True, -1), "sequences": (3, True, -1), "4bit": (4, True, -1),
         "fasta_nomask": (0, False, -1), "fasta_ll13": (0, True, 13), "fasta_ll0": (0, True, 0), "fastq": (1, True, -1)}


@pytest.mark.parametrize("case", naf_cases(), ids=lambda c: c["name"])
def test_oracle_unnaf_matches_reference_outputs(oracle, case):
    naf = golden_bytes("naf", case["name"] + ".naf")
    assert len(naf) == case["naf_len"]
    for m, (mode, use_mask, ll) in MODES.items():
        if m not in case["outputs"]:
            continue
        out = oracle.unnaf(naf, mode, use_mask=use_mask, line_length=ll)
        assert len(out) == case["outputs"][m]["len"], m
        assert sha(out) == case["outputs"][m]["sha256"], m


@pytest.mark.parametrize("case", naf_cases(), ids=lambda c: c["name"])
def test_oracle_split_matches_reference_streams(oracle, case):
    """ennaf direction: the six streams the oracle derives from the text == the streams inside the
    reference-made archive (decoded with the oracle's zstd)."""
    O = oracle
    naf = golden_bytes("naf", case["name"] + ".naf")
    h = O.parse_naf(naf)
    args = case["ennaf_args"]
    st = _seq_type(args, O)
    try:
        text = golden_bytes("naf", case["name"] + ".in")
    except FileNotFoundError:
        if h.flags & 1:
            text = O.unnaf(naf, O.MODE_FASTQ)         # FASTQ text minus the mask (unnaf.c:442)
        else:
            text = O.unnaf(naf, O.MODE_FASTA)
    sp = O.split_text(text, st, "--no-mask" in args)
    streams = [sp.ids, sp.comments, sp.lengths, sp.mask, sp.seq, sp.qual]
    for i in range(6):
        if h.payload_off[i] is None:
            continue
        if i == O.MASK and (h.flags & 1) and not golden_exists(case):
            continue                                   # mask is lost in the FASTQ round trip
        assert O.zstd_decompress(h.frame(naf, i)) == streams[i], i
    if "--line-length" not in args:
        assert h.line_length == sp.longest_line
    assert h.n_sequences == sp.n_sequences and h.orig[O.SEQ] == sp.n_bases
    # and the oracle's own archive decodes to the same text under the oracle
    naf2 = O.ennaf(text, st, "--no-mask" in args)
    assert O.unnaf(naf2, -1) == O.unnaf(naf, -1) or "--line-length" in args


def golden_exists(case):
    import os
    from conftest import GOLDEN
    return os.path.exists(os.path.join(GOLDEN, "naf", case["name"] + ".in"))


# ---- zstd frames made by libzstd --------------------------------------------------------------------
@pytest.mark.parametrize("case", zstd_cases(), ids=lambda c: c["name"])
def test_oracle_zstd_matches_libzstd_frames(oracle, case):
    frame = golden_bytes("zstd", case["name"] + ".zst")
    out = oracle.zstd_decompress(frame)
    assert len(out) == case["len"] and sha(out) == case["sha256"]


def test_golden_frames_cover_every_feature_class():
    """the decoder must be proven on every literal type and sequence mode."""
    lit = np.zeros(4, dtype=int)
    modes = np.zeros((3, 4), dtype=int)
    wlogs = set()
    infos = [c["first_frame"] for c in zstd_cases() if c["first_frame"]]
    for c in naf_cases():
        infos += list(c["frame_info"].values())
    for fi in infos:
        lit += np.array(fi["lit"])
        modes += np.array(fi["modes"])
        wlogs.add(fi["wlog"])
    assert lit[0] > 0 and lit[2] > 0 and lit[3] > 0           # raw, huffman, treeless
    assert (modes > 0).all(axis=None) or (modes[:, [0, 2, 3]] > 0).all()   # predefined / fse / repeat on LL, OF, ML
    assert modes[1][1] > 0 and modes[2][1] > 0                 # rle mode seen on OF and ML
    assert {19, 23, 27} <= wlogs


def test_oracle_raw_store_roundtrip(oracle):
    for n in (0, 1, 131072, 131073, 400000):
        d = bytes(np.random.default_rng(n).integers(0, 256, n, dtype=np.uint8))
        assert oracle.zstd_decompress(oracle.zstd_store_raw(d)) == d


# ---- building blocks -----------------------------------------------------------------------------------
def test_mask_rle_units_and_inverse(oracle):
    cases = {b"ACGT": b"\x04", b"acgt": b"\x00\x04", b"ACgt": b"\x02\x02", b"": b"",
             b"A" * 255: b"\xff\x00", b"A" * 254: b"\xfe", b"A" * 256: b"\xff\x01",
             b"a" * 510 + b"C": b"\x00\xff\xff\x00\x01"}
    for s, units in cases.items():
        assert oracle.mask_rle(s) == units, s[:10]
        assert oracle.mask_apply(s.upper(), units) == s
    rng = np.random.default_rng(0)
    s = bytearray()
    while len(s) < 200000:
        run = int(rng.choice([1, 2, 100, 254, 255, 256, 509, 510, 511, 1000]))
        s += (b"acgt" if rng.random() < 0.5 else b"ACGT")[int(rng.integers(0, 4)):][:1] * run
    s = bytes(s)
    assert oracle.mask_apply(s.upper(), oracle.mask_rle(s)) == s


def test_fastq_error_messages(oracle):
    with pytest.raises(ValueError, match="doesn't match sequence length"):
        oracle.split_text(b"@r1\nACGT\n+\n!!!\n")
    with pytest.raises(ValueError, match="last sequence has no quality"):
        oracle.split_text(b"@r1\nACGT\n")
    with pytest.raises(ValueError, match="neither '>' nor '@'"):
        oracle.split_text(b"hello")
    with pytest.raises(ValueError, match="not at the beginning of the line"):
        oracle.split_text(b" >x\nAC")
    assert oracle.split_text(b"").format == oracle.FMT_UNKNOWN


@pytest.mark.skipif(not __import__("oracle.oracle", fromlist=["x"]).have_ref(), reason="oracle/_ref not built")
def test_oracle_against_live_reference_fuzz(oracle):
    """Only where oracle/_ref exists (build container, GPU box): random malformed-ish FASTA."""
    O = oracle
    rng = np.random.default_rng(11)
    alphabet = np.frombuffer(b">>\n\n\r\t ACGTNacgtn-XZ*\x00\x7f\xff\x0b", dtype=np.uint8)
    for i in range(40):
        t = b">" + alphabet[rng.integers(0, len(alphabet), int(rng.integers(1, 300)))].tobytes()
        naf = O.ref_ennaf(t)
        h = O.parse_naf(naf)
        sp = O.split_text(t)
        for k, s in enumerate([sp.ids, sp.comments, sp.lengths, sp.mask, sp.seq]):
            assert O.zstd_decompress(h.frame(naf, k)) == s
        if h.n_sequences and h.orig[O.SEQ] == sum(np.frombuffer(sp.lengths, dtype="<u4").astype(int)):
            assert O.ref_unnaf(naf, ("--fasta",)) == O.unnaf(naf, O.MODE_FASTA)


def test_cfg1_ten_megabases_cpu_round_trip(oracle):
    """BASELINE configs[0] / SURVEY 8(d) cfg1: one record `>seq1 synthetic random ACGT`, 10 000 000 uniform bases (PCG64 seed
    12345), 80-column lines = 10 125 028 bytes; the reference's archive of it was measured at 2 500 852 bytes (2.0007 bit / base).
    The CPU plumbing both ways: the restatement's archive decodes under the real reference, the reference's under the
    restatement, and both decode their own."""
    from naf_amd import synth
    text = synth.fasta_acgt(10_000_000, 1, 80, seed=12345)
    assert len(text) == 10_125_028 and text.startswith(b">seq1 synthetic random ACGT\n")
    mine = oracle.ennaf(text)                                      # (the restatement stores raw blocks: its archive is the 4-bit stream, 5 MB)
    assert oracle.unnaf(mine, 0) == text
    h = oracle.parse_naf(mine)
    assert h.orig[4] == 10_000_000 and h.n_sequences == 1 and h.line_length == 80
    if oracle.have_ref():
        ref = oracle.ref_ennaf(text)
        assert abs(len(ref) - 2_500_852) <= 64, len(ref)           # libzstd's level-1 entropy coding of 16 equally likely pair codes
        assert oracle.unnaf(ref, 0) == text
        assert oracle.ref_unnaf(ref) == text
        assert oracle.ref_unnaf(mine) == text
