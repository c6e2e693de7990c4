"""CPU-only tests of the host side: C-ABI surface, kernel logic single-stepped on the host, the
CLI paths that need no device, failing loudly without a GPU, and the multi-rank plumbing on gloo."""
import ctypes
import hashlib
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

SEED = int(os.environ.get("NAF_TEST_SEED", "0"))          # other texts of the same kinds (tests/test_gpu_encode.py)

from conftest import GOLDEN, ROOT, golden_bytes, naf_cases, zstd_cases

BIN = os.path.join(ROOT, "naf_amd", "bin")


def sha(b):
    return hashlib.sha256(b).hexdigest()


@pytest.fixture(scope="session", autouse=True)
def built():
    if not (os.path.exists(os.path.join(ROOT, "naf_amd", "libnaf_gpu.so")) and os.path.exists(os.path.join(BIN, "unnaf"))
            and os.path.exists(os.path.join(ROOT, "tests", "emul", "libzstd_emul.so"))):
        env = dict(os.environ, PATH="/opt/rocm/bin:" + os.environ.get("PATH", ""))
        subprocess.check_call(["make", "-s", "-j8", "-C", ROOT, "all"], env=env)


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "naf_gpu.h")).read()
    declared = sorted(set(re.findall(r"\b(naf_gpu_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 20
    lib = ctypes.CDLL(os.path.join(ROOT, "naf_amd", "libnaf_gpu.so"))      # loads without a GPU; no compute calls
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    from naf_amd import capi
    assert sorted(capi.EXPORTS) == declared


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = ctypes.CDLL(os.path.join(ROOT, "naf_amd", "libnaf_gpu.so"))
    h = ctypes.c_void_p()
    assert lib.naf_gpu_init(0, ctypes.byref(h)) == -1          # NAF_GPU_ENODEV, and no context
    assert not h.value
    naf = os.path.join(GOLDEN, "naf", "acgt_10k.naf")
    p = subprocess.run([os.path.join(BIN, "unnaf"), "--fasta", "-c", naf], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and p.stdout == b"" and b"unnaf error: can't initialize the GPU path" in p.stderr
    p = subprocess.run([os.path.join(BIN, "ennaf"), "-c"], input=b">a\nACGT\n", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert p.returncode == 1 and p.stdout == b"" and b"ennaf error: can't initialize the GPU path" in p.stderr


def test_product_never_imports_the_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "naf_amd")):
        for f in fs:
            if f.endswith((".py", ".c", ".h", ".hip")):
                src = open(os.path.join(dp, f), errors="replace").read()
                assert "oracle" not in src.replace("oracle/_ref", "").lower() or f == "shard.py" and False, os.path.join(dp, f)


# ---- kernel logic single-stepped on the host (tests/emul) ---------------------------------------------------
@pytest.fixture(scope="module")
def emul():
    L = ctypes.CDLL(os.path.join(ROOT, "tests", "emul", "libzstd_emul.so"))
    L.emul_zstd_decompress_frame.restype = ctypes.c_longlong
    L.emul_zstd_decompress_frame.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    L.emul_zstd_compress.restype = ctypes.c_longlong
    L.emul_zstd_compress.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t]
    return L


@pytest.mark.parametrize("case", [c for c in zstd_cases() if c["name"] not in ("two_frames", "skippable_then_frame")], ids=lambda c: c["name"])
def test_decoder_kernel_logic_on_golden_frames(emul, case):
    frame = golden_bytes("zstd", case["name"] + ".zst")
    out = ctypes.create_string_buffer(case["len"] + 64)
    n = emul.emul_zstd_decompress_frame(frame, len(frame), out, case["len"] + 64)
    assert n == case["len"] and sha(out.raw[:n]) == case["sha256"]


def test_decoder_kernel_logic_on_reference_archives(emul, oracle):
    for case in naf_cases():
        naf = golden_bytes("naf", case["name"] + ".naf")
        h = oracle.parse_naf(naf)
        for i in range(6):
            if h.payload_off[i] is None:
                continue
            f = h.frame(naf, i)
            ref = oracle.zstd_decompress(f)
            out = ctypes.create_string_buffer(len(ref) + 64)
            n = emul.emul_zstd_decompress_frame(f, len(f), out, len(ref) + 64)
            assert n == len(ref) and out.raw[:n] == ref, (case["name"], i)


def test_encoder_kernel_logic_roundtrips_through_oracle(emul, oracle):
    rng = np.random.default_rng(3 + SEED)
    syms = np.array([0x88, 0x84, 0x82, 0x81, 0x48, 0x44, 0x42, 0x41, 0x28, 0x24, 0x22, 0x21, 0x18, 0x14, 0x12, 0x11], dtype=np.uint8)
    p2 = np.array([2.0 ** -(i + 1) for i in range(40)])
    data = [b"", b"A", b"\x07" * 100000, syms[rng.integers(0, 16, 300001)].tobytes(),
            b"".join(b"read%d len=%d\x00" % (i, 100 + i % 50) for i in range(5000)),
            rng.integers(33, 74, 100000, dtype=np.uint8).tobytes(), rng.integers(0, 256, 50000, dtype=np.uint8).tobytes(),
            rng.choice(np.arange(40, dtype=np.uint8) + 60, 200000, p=p2 / p2.sum()).tobytes()]
    data += [rng.integers(65, 70, n, dtype=np.uint8).tobytes() for n in (2, 3, 63, 64, 65, 255, 1000)]
    for d in data:
        for blk in (1024, 32768, 131072):
            cap = len(d) + len(d) // 64 + 1024
            out = ctypes.create_string_buffer(cap)
            n = emul.emul_zstd_compress(d, len(d), blk, out, cap)
            assert n > 0
            assert oracle.zstd_decompress(out.raw[:n], len(d) + 16) == d


def test_lz_block_format_against_three_decoders(emul, oracle):
    """The LZ-coded block layout of the GPU encoder (sequences with predefined FSE tables, new-offset codes only, literals
    Huffman / raw / RLE), produced here by a serial reference of the same layout: decodable by the from-spec oracle, by the
    decoder kernels' logic (emul) and -- when this machine has it -- by libzstd itself."""
    emul.emul_zstd_compress_lz.restype = ctypes.c_longlong
    emul.emul_zstd_compress_lz.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t]
    zlib = None
    for cand in ("/opt/conda/lib/libzstd.so", "libzstd.so.1"):
        try:
            zlib = ctypes.CDLL(cand); break
        except OSError:
            pass
    if zlib is not None:
        zlib.ZSTD_decompress.restype = ctypes.c_size_t
        zlib.ZSTD_decompress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t]
    rng = np.random.default_rng(21 + SEED)
    data = [b"", b"A", b"abcabcabcabcabc", b"".join(b"SRR%07d.%d length=%d\x00" % (1234567, i, 150) for i in range(1, 6000)),
            np.full(20000, 150, dtype="<u4").tobytes(), (b"abcdefghij" * 700 + rng.integers(0, 256, 3000, dtype=np.uint8).tobytes()) * 4,
            b"A" * 40000 + b"CGT" * 9000, rng.integers(0, 4, 50000, dtype=np.uint8).tobytes()]
    for i in range(40):
        n = int(rng.integers(1, 30000)); a = int(rng.choice([2, 4, 16, 256]))
        d = rng.integers(0, a, n, dtype=np.uint8).tobytes()
        data.append(d[: n // 3] * 3 if i % 2 else d)
    for d in data:
        for blk in (256, 4096, 32768):
            cap = 2 * len(d) + 1024
            out = ctypes.create_string_buffer(cap)
            n = emul.emul_zstd_compress_lz(d, len(d), blk, out, cap)
            assert n > 0, (len(d), blk, n)
            frame = out.raw[:n]
            assert oracle.zstd_decompress(frame, len(d) + 16) == d
            back = ctypes.create_string_buffer(len(d) + 64)
            assert emul.emul_zstd_decompress_frame(frame, n, back, len(d) + 64) == len(d) and back.raw[:len(d)] == d
            if zlib is not None:
                r = zlib.ZSTD_decompress(back, len(d) + 64, frame, n)
                assert r == len(d) and back.raw[:r] == d


def test_lzx_block_format_against_the_decoders(emul, oracle):
    """The cross-block stage of the GPU encoder (zstd_enc.hip: k_ldm_insert / k_lzx_parse / k_lzx_seqenc) as a serial model that shares
    its sequence writer (zenc_write_sequences_x: repeat-offset codes from a per-block history, predefined / RLE / FSE_Compressed
    tables chosen per block): frames decode under the from-spec oracle and the decoder kernels' logic, windows 2^10 .. 2^27."""
    emul.emul_zstd_compress_lzx.restype = ctypes.c_longlong
    emul.emul_zstd_compress_lzx.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32)]
    rng = np.random.default_rng(23 + SEED)
    unit = rng.integers(0, 16, 30000, dtype=np.uint8).tobytes()
    def mutated(k):
        a = bytearray(unit)
        for i in rng.integers(0, len(a), k):
            a[i] = int(rng.integers(0, 16))
        return bytes(a)
    data = [b"", b"A", b"abcabcabcabcabc" * 3, b"".join(b"SRR%07d.%d length=%d\x00" % (1234567, i, 150) for i in range(1, 9000)),
            np.full(40000, 150, dtype="<u4").tobytes(), b"".join(mutated(40) + rng.integers(0, 16, int(rng.integers(1, 999)), dtype=np.uint8).tobytes() for _ in range(8)),
            b"A" * 70000 + b"CGT" * 30000 + b"A" * 70000, rng.integers(0, 4, 90000, dtype=np.uint8).tobytes(),
            unit[:5000] + b"x" + unit[:5000] + b"yz" + unit[:5000] + b"abc" + unit[:5000] + b"defg" + unit[:5000]]
    for i in range(12):
        n = int(rng.integers(1, 60000)); a = int(rng.choice([2, 4, 16, 256]))
        d = rng.integers(0, a, n, dtype=np.uint8).tobytes()
        data.append(d[: n // 3] * 3 if i % 2 else d)
    st = (ctypes.c_uint32 * 3)()
    reps = 0
    for d in data:
        for blk, wlog in ((4096, 10), (16384, 17), (32768, 20), (65535, 27)):
            cap = 2 * len(d) + 4096
            out = ctypes.create_string_buffer(cap)
            n = emul.emul_zstd_compress_lzx(d, len(d), blk, wlog, out, cap, st)
            assert n > 0, (len(d), blk, wlog, n)
            reps += st[1]
            frame = out.raw[:n]
            assert oracle.zstd_decompress(frame, len(d) + 16) == d, (len(d), blk, wlog)
            back = ctypes.create_string_buffer(len(d) + 64)
            assert emul.emul_zstd_decompress_frame(frame, n, back, len(d) + 64) == len(d) and back.raw[:len(d)] == d
    assert reps > 1000                                          # repeat-offset codes are exercised


def test_lzx_model_on_the_repeat_genomes_is_close_to_the_reference(emul, oracle):
    """on the repeat-rich golden inputs the cross-block stage stays within 10 % of the reference's archive made
    with the same flags (tests/golden/naf/repeat_*.naf, written by the real ennaf -19 / -3 --long 27)."""
    from naf_amd import synth as mg
    emul.emul_zstd_compress_lzx.restype = ctypes.c_longlong
    emul.emul_zstd_compress_lzx.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint32)]
    for name, text, wlog in (("repeat_l19", mg.repeat_genome(), 23), ("repeat_long27", mg.repeat_genome(seed=11, unit=300000, copies=8), 27)):
        ref = open(os.path.join(ROOT, "tests", "golden", "naf", name + ".naf"), "rb").read()
        seq = oracle.split_text(text, 0, False).seq
        assert oracle.zstd_decompress(oracle.parse_naf(ref).frame(ref, 4), len(seq) + 16) == seq     # the same stream the reference compressed
        cap = len(seq) + 4096
        out = ctypes.create_string_buffer(cap)
        n = emul.emul_zstd_compress_lzx(seq, len(seq), 65535, wlog, out, cap, None)
        assert n > 0 and oracle.zstd_decompress(out.raw[:n], len(seq) + 16) == seq
        ref_frame = oracle.parse_naf(ref).comp[4]
        assert n <= 1.10 * ref_frame, (name, n, ref_frame)


def test_decoder_window_reader_at_every_rate_and_alignment(emul):
    """k_huf_literals' sector-window reader (two-sector LDS ring, register-staged prefetch; four sectors above 7-bit codes),
    single-stepped on the host: streams from 1 bit to 11 bits per symbol, every start alignment class, against the plain reader.
    (A 1-bit code moves the read position by 4 bytes a round -- the case that once let the ring overwrite live bytes.)"""
    emul.emul_window_roundtrip.restype = ctypes.c_int
    emul.emul_window_roundtrip.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int]
    rng = np.random.default_rng(11 + SEED)
    streams = []
    for alpha in (2, 3, 4, 16, 41, 100, 256):
        streams.append(rng.integers(0, alpha, 9000, dtype=np.uint8).tobytes())
    p2 = np.array([2.0 ** -(i + 1) for i in range(30)])
    streams.append(rng.choice(np.arange(30, dtype=np.uint8), 9000, p=p2 / p2.sum()).tobytes())       # codes up to 11 bits
    streams.append((b"\x00" * 50 + b"\x01") * 170)                                                    # 1-bit code, long runs
    for d in streams:
        for n in (len(d), 6009, 999, 257):
            for align in (0, 1, 7, 8, 22, 37, 55, 56, 57, 63):
                r_big = emul.emul_window_roundtrip(d[:n], n, align, 1)
                assert r_big == 0, (len(set(d)), n, align, "big", r_big)
                r = emul.emul_window_roundtrip(d[:n], n, align, 0)
                assert r in (0, -5), (len(set(d)), n, align, "small", r)                           # -5: table wider than 7 bits


def test_huffman_stream_decoded_in_parts(emul):
    """k_huf_par's algorithm single-stepped on the host (zstd_dec_core.h: hufw_*, hufp_*): a stream cut into P parts, every part
    started inside its predecessor, re-walked until every start equals its predecessor's end, then decoded part by part -- against
    the serial reader.  Alphabets from 1-bit to 11-bit codes, nearly flat trees (which fall into step slowly: rounds of re-walking
    must happen and must converge), every P, margins down to none at all, streams shorter than their number of parts."""
    emul.emul_parts_roundtrip.restype = ctypes.c_int
    emul.emul_parts_roundtrip.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint32)]
    rng = np.random.default_rng(12 + SEED)
    streams = []
    for alpha in (2, 3, 4, 16, 17, 41, 100, 256):
        streams.append(rng.integers(0, alpha, 9000, dtype=np.uint8).tobytes())
    p2 = np.array([2.0 ** -(i + 1) for i in range(30)])
    streams.append(rng.choice(np.arange(30, dtype=np.uint8), 9000, p=p2 / p2.sum()).tobytes())       # codes up to 11 bits
    streams.append((b"\x00" * 50 + b"\x01") * 170)                                                    # 1-bit code, long runs
    pr = np.array([1.0] * 15 + [0.5, 0.5]); streams.append(rng.choice(np.arange(17, dtype=np.uint8), 33000, p=pr / pr.sum()).tobytes())   # fifteen 4-bit and two 5-bit codes
    pq = np.array([.0826, .0826, .0854, .0574, .0826, .0574, .0126, .0604, .0604, .042, .042, .0574, .0854, .0604, .0604, .0826, 1e-4, 1e-5])
    streams.append(rng.choice(np.arange(18, dtype=np.uint8), 33000, p=pq / pq.sum()).tobytes())       # pairs of a GC-poor genome and two rare codes
    rounds = ctypes.c_uint32(0)
    saw_rounds = 0
    for d in streams:
        for n in (len(d), 6009, 999, 257, 40, 3):
            for P in (1, 2, 4, 16, 64):
                for margin in (0, 256, 64, 8):
                    for align in (0, 3, 61, (1 << 32) | 5):        # (bit 32: the stream is all of the readable buffer)
                        r = emul.emul_parts_roundtrip(d[:n], n, P, margin, align, ctypes.byref(rounds))
                        assert r in (0, -10), (len(set(d)), n, P, margin, align, r)             # -10: a single distinct symbol has no Huffman stream
                        saw_rounds += rounds.value
    assert saw_rounds > 100                                      # the re-walk rounds did run (tiny margins, nearly flat trees)
    # with the kernel's own margin the common alphabets fall into step at once nearly always
    d = streams[-1]
    tot = 0
    for P in (16, 64):
        for k in range(20):
            assert emul.emul_parts_roundtrip(d[k * 100:], len(d) - k * 100, P, 0, 0, ctypes.byref(rounds)) == 0
            tot += rounds.value
    assert tot <= 6                                              # (one stream in forty, measured)


# ---- CLI paths that need no device ---------------------------------------------------------------------------------
def run(prog, *args, stdin=None):
    p = subprocess.run([os.path.join(BIN, prog), *args], input=stdin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
    return p.returncode, p.stdout, p.stderr


def test_cli_interface_fixtures_of_the_reference():
    """The reference's tests/interface set: `--version`, and a start without arguments on a terminal (exit code 0, the hint on
    stderr, nothing on stdout -- ennaf.c:439-443, unnaf.c:364-368).  The terminal is a pty."""
    import pty
    for prog in ("ennaf", "unnaf"):
        rc, out, err = run(prog, "--version")
        assert rc == 0
        assert out == golden_bytes("ref_tests", "interface", prog + "-version.out-ref")
        assert err == golden_bytes("ref_tests", "interface", prog + "-version.err-ref")
        master, slave = pty.openpty()
        try:
            p = subprocess.run([os.path.join(BIN, prog)], stdin=slave, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=60)
        finally:
            os.close(master); os.close(slave)
        assert p.returncode == 0
        assert p.stdout == golden_bytes("ref_tests", "interface", prog + "-no-input.out-ref")
        assert p.stderr == golden_bytes("ref_tests", "interface", prog + "-no-input.err-ref")
    rc, out, err = run("unnaf", "--bogus")
    assert rc == 1 and err == b'unnaf error: unknown or incomplete argument "--bogus"\n'
    rc, out, err = run("ennaf", "-c", "-o", "x")
    assert rc == 1 and err == b"ennaf error: '-c' and '-o' can't be used together\n"


def test_cli_header_only_modes(oracle):
    for case in naf_cases():
        path = os.path.join(GOLDEN, "naf", case["name"] + ".naf")
        naf = golden_bytes("naf", case["name"] + ".naf")
        h = oracle.parse_naf(naf)
        rc, out, err = run("unnaf", "--number", path)
        assert (rc, out, err) == (0, b"%d\n" % h.n_sequences, b"")
        rc, out, err = run("unnaf", "--format", path)
        tn = ["DNA", "RNA", "protein", "text"][h.seq_type]
        assert out == ("%s sequences%s in NAF format version %d\n" % (tn, " with qualities" if h.flags & 1 else "", h.version)).encode()
        rc, out, err = run("unnaf", "--total-length", path)
        assert out == b"%d\n" % h.orig[4]
        rc, out, err = run("unnaf", "--sizes", path)
        assert out.startswith(b"IDs: %d / %d" % (h.comp[0], h.orig[0])) or b"Title" in out
    rc, out, err = run("unnaf", "--title", os.path.join(GOLDEN, "naf", "title.naf"))
    assert out == b"my title\n"
    rc, out, err = run("unnaf", "--number", stdin=b"garbage!")
    assert rc == 1 and err == b"unnaf error: not a NAF format\n"


# ---- multi-rank plumbing on gloo (world_size 2) ---------------------------------------------------------------------
def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from naf_amd import shard
    total = 1_000_003
    whole = (torch.arange(total, dtype=torch.int64) * 7 % 251).to(torch.uint8)
    b, e = shard.byte_range(total, rank, world)
    got = shard.gather_ranges(whole[b:e].clone(), total, dst=0)
    ok = True
    if rank == 0:
        ok = bool(torch.equal(got, whole))
    t = torch.tensor([1.0 + rank]); dist.all_reduce(t, op=dist.ReduceOp.MAX)     # the bench's max-over-ranks timing reduction
    ok = ok and float(t.item()) == float(world)
    q.put((rank, ok, b, e))
    dist.destroy_process_group()


def test_sharded_gather_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    ps = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = sorted(q.get(timeout=180) for _ in ps)
    for p in ps:
        p.join(60)
    assert all(r[1] for r in res)
    assert res[0][2] == 0 and res[0][3] == res[1][2] and res[1][3] == 1_000_003      # ranges tile the text


def test_byte_ranges_tile_any_total():
    from naf_amd import shard
    for total in (0, 1, 4095, 4096, 4097, 10**9 + 7):
        for world in (1, 2, 3, 8):
            pos = 0
            for r in range(world):
                b, e = shard.byte_range(total, r, world)
                assert b == min(pos, total) and e >= b
                pos = e
            assert pos == total


# ---- encoder front end: the four-bytes-per-instruction byte classes against the plain predicates -------------------------------
def _quick_table(expected):
    """host side of enc.hip:set_expected -- slot (c >> 1) & 7 holds the accepted upper-case letter, 0xFF elsewhere"""
    t = [0xFF] * 8
    for ch in b"ACGTUN":
        if ch in expected and (ch | 0x20) in expected and t[(ch >> 1) & 7] == 0xFF:
            t[(ch >> 1) & 7] = ch
    return int.from_bytes(bytes(t[:4]), "little"), int.from_bytes(bytes(t[4:]), "little"), {c for c in t if c != 0xFF}


@pytest.mark.parametrize("alphabet", [b"ABCDGHKMNRSTVWY", b"ABCDGHKMNRSUVWY"])
def test_swar_piece_plain_every_byte_every_position(alphabet):
    """enc_swar.h piece_plain: the test behind the pure-tile paths of k_enc_count / k_enc_scatter -- quick letters, LF and CR only."""
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "emul", "libzstd_emul.so"))
    expected = set(alphabet) | {c | 0x20 for c in alphabet} | {ord("-")}
    qlo, qhi, quick = _quick_table(expected)
    t = bytearray(qlo.to_bytes(4, "little") + qhi.to_bytes(4, "little"))
    assert t[5] == 0xFF and t[6] == 0xFF                        # the two slots the line ends go into are free
    t[5], t[6] = 0x0A, 0x0D
    plo, phi = int.from_bytes(bytes(t[:4]), "little"), int.from_bytes(bytes(t[4:]), "little")
    rng = np.random.default_rng(6 + SEED)
    eol = ctypes.c_uint32()
    fills = [bytes([65] * 16), bytes([10] * 16), bytes([13] * 16), bytes(rng.choice(np.frombuffer(b"ACGTNacgtnUu\n\r", dtype=np.uint8), 16)),
             bytes(rng.choice(np.frombuffer(b"ACGTNacgtn\n", dtype=np.uint8), 16))]
    ok = lambda c: c in (0x0A, 0x0D) or ((c & 0xDF) in quick and c in expected and c >= 0x40)
    for fill in fills:
        for pos in range(16):
            for b in range(256):
                p = bytearray(fill); p[pos] = b
                got = lib.emul_piece_plain(bytes(p), plo, phi, ctypes.byref(eol))
                assert got == int(all(ok(c) for c in p)), (fill, pos, b)
                if got:
                    assert eol.value == sum(1 << i for i, c in enumerate(p) if c in (0x0A, 0x0D)), (fill, pos, b)


@pytest.mark.parametrize("alphabet", [b"ABCDGHKMNRSTVWY", b"ABCDGHKMNRSUVWY", bytes(range(0x41, 0x5B))])
def test_swar_piece_flags_every_byte_every_position(alphabet):
    lib = ctypes.CDLL(os.path.join(ROOT, "tests", "emul", "libzstd_emul.so"))
    expected = set(alphabet) | {c | 0x20 for c in alphabet} | {ord("-")}
    qlo, qhi, quick = _quick_table(expected)
    assert quick                                                # the table is not empty for any of the reference's alphabets
    rng = np.random.default_rng(5 + SEED)
    out = (ctypes.c_uint32 * 8)()
    fills = [bytes([65] * 16), bytes([10] * 16), bytes([0x20] * 16), bytes([0x7E] * 16), bytes(rng.integers(0, 256, 16, dtype=np.uint8)),
             bytes(rng.choice(np.frombuffer(b"ACGTNacgtn\n", dtype=np.uint8), 16))]
    is_eol = lambda c: 0x0A <= c <= 0x0D
    is_sp = lambda c: 0x09 <= c <= 0x0D or c == 0x20
    is_quick = lambda c: (c & 0xDF) in quick and c in expected
    for fill in fills:
        for pos in range(16):
            for b in range(256):
                p = bytearray(fill); p[pos] = b
                lib.emul_piece_flags(bytes(p), qlo, qhi, out)
                assert out[0] == sum(1 << i for i, c in enumerate(p) if is_eol(c)), (fill, pos, b)
                assert out[1] == sum(1 << i for i, c in enumerate(p) if is_sp(c)), (fill, pos, b)
                assert out[2] == sum(1 << i for i, c in enumerate(p) if c == 0x3E), (fill, pos, b)
                assert out[3] == int(all(is_sp(c) or is_quick(c) for c in p)), (fill, pos, b)
                assert out[4] == int(all(is_quick(c) for c in p)), (fill, pos, b)
                assert out[5] == int(all(0x21 <= c <= 0x7E for c in p)), (fill, pos, b)
                assert out[6] == sum(1 << i for i, c in enumerate(p) if c < 0x20 or c in (0x7F, 0xFF)), (fill, pos, b)
                assert out[7] == sum(1 << i for i, c in enumerate(p) if not 0x21 <= c <= 0x7E), (fill, pos, b)
                # the quick test is only ever a sufficient one
                if out[3]: assert all(is_sp(c) or c in expected for c in p)
