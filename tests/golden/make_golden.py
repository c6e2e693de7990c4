#!/usr/bin/env python3
"""Regenerates tests/golden/ -- run in the build container only (needs /root/reference + libzstd).

What it writes (all DATA: inputs, reference-made archives, expected outputs / their SHA-256):
  ref_tests/<set>/...      the reference's own test fixtures (inputs, *.out-ref, *.err-ref) and
                           ref_cases.json = the (ennaf args, unnaf args) of each reference *.test
  naf/<case>.naf           archives made by the REAL reference ennaf (oracle/_ref/ennaf)
  naf/<case>.in            the input text (only when small)
  naf_cases.json           per case: ennaf args, sizes and SHA-256 of reference unnaf outputs per mode
  zstd/<case>.zst          frames made by the image's libzstd 1.4.9 (library + CLI)
  zstd_cases.json          per frame: raw length + SHA-256 of the content, and its feature class
Nothing here is imported by the product.
"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import oracle as O          # noqa: E402
from naf_amd import synth                # noqa: E402

REF = "/root/reference"
ZSTD_CLI = "/opt/conda/bin/zstd"


def sha(b):
    return hashlib.sha256(b).hexdigest()


def copy_ref_tests():
    cases = []
    for tset in ("alphabet", "charcount", "small", "large"):
        src = os.path.join(REF, "tests", tset)
        dst = os.path.join(HERE, "ref_tests", tset)
        os.makedirs(dst, exist_ok=True)
        for f in sorted(os.listdir(src)):
            if f.endswith(".fa") or f.endswith("-ref"):
                shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
            if f.endswith(".test"):
                line = open(os.path.join(src, f)).read().strip()
                m = re.match(r"ennaf (.*?)\{GROUP\}\.fa 2>\{TEST\}\.e\.err \| unnaf (.*?)>\{TEST\}\.out 2>\{TEST\}\.u\.err$", line)
                assert m, line
                name = f[:-5]
                cases.append({"set": tset, "name": name, "input": name.split("-")[0] + ".fa",
                              "ennaf_args": m.group(1).split(), "unnaf_args": m.group(2).split()})
    # the interface set: expected stdout / stderr of `--version` and of a start without input on a terminal (data only; the tests
    # build the command lines themselves)
    src = os.path.join(REF, "tests", "interface")
    dst = os.path.join(HERE, "ref_tests", "interface")
    os.makedirs(dst, exist_ok=True)
    for f in sorted(os.listdir(src)):
        if f.endswith("-ref"):
            shutil.copyfile(os.path.join(src, f), os.path.join(dst, f))
    json.dump(cases, open(os.path.join(HERE, "ref_cases.json"), "w"), indent=1)
    print("ref cases:", len(cases))


def mask_boundary_fasta():
    parts = []
    for k, run in enumerate((254, 255, 256, 510, 509, 511, 765, 1)):
        seq = (b"A" * run + b"c" * run + b"G" * 3 + b"t" * run + b"N" * run)
        parts.append(b">m%d run=%d\n" % (k, run) + synth.wrap_lines(np.frombuffer(seq, dtype=np.uint8), 70))
    parts.append(b">startmasked\nacgtACGTacgt\n")
    return b"".join(parts)


repeat_genome = synth.repeat_genome


def naf_cases():
    os.makedirs(os.path.join(HERE, "naf"), exist_ok=True)
    tiny_many = b"".join(b">t%d\n%s\n" % (i, b"ACGTN"[: 1 + i % 5] * (1 + i % 6)) for i in range(3000))
    cases = [
        ("acgt_10k", synth.fasta_acgt(10000, 1, 80), []),
        ("acgt_odd", synth.fasta_acgt(100001, 2, 80, seed=9), []),
        ("acgt_1m2", synth.fasta_acgt(1200000, 3, 80, seed=4), []),
        ("acgt_ll1", synth.fasta_acgt(5000, 2, 1, seed=5), []),
        ("acgt_nowrap", synth.fasta_acgt(200000, 2, 0, seed=6), []),
        ("mixed_60", synth.fasta_mixed(40, 3000, 60, 1), []),
        ("mixed_nomask", synth.fasta_mixed(12, 2000, 50, 8), ["--no-mask"]),
        ("mask_bounds", mask_boundary_fasta(), []),
        ("tiny_many", tiny_many, []),
        ("repeat_l1", repeat_genome(), []),
        ("repeat_l19", repeat_genome(), ["-19"]),
        ("repeat_long27", repeat_genome(seed=11, unit=300000, copies=8), ["-3", "--long", "27"]),
        ("fastq_4k", synth.fastq_reads(4000, 150), []),
        ("fastq_var", synth.fastq_reads(1500, 120, seed=3, var_len=True), []),
        ("rna_small", synth.fasta_mixed(6, 500, 60, 4).replace(b"T", b"U").replace(b"t", b"u"), ["--rna"]),
        ("protein_small", b">p1 prot\nMKVLAAGIVGLLLAQWERTYIPASDFGHKLCVNM*\n>p2\nmkvl-xbzj\n", ["--protein"]),
        ("text_small", b">t1 text\nHello,World!123\n>t2\n<<>>[]{}\n", ["--text"]),
        ("crlf", b">a b\r\nACGT\r\nAC\r\n\r\n>b\r\n\r\nGG\r\n", []),
        ("ll_override", synth.fasta_acgt(3000, 2, 80, seed=12), ["--line-length", "37"]),
        ("title", synth.fasta_acgt(500, 1, 80, seed=13), ["--title", "my title"]),
    ]
    modes = {"fasta": ["--fasta"], "seq": ["--seq"], "sequences": ["--sequences"], "4bit": ["--4bit"],
             "fasta_nomask": ["--fasta", "--no-mask"], "fasta_ll13": ["--fasta", "--line-length", "13"],
             "fasta_ll0": ["--fasta", "--line-length", "0"], "fastq": ["--fastq"],
             "ids": ["--ids"], "names": ["--names"], "lengths": ["--lengths"], "mask": ["--mask"]}
    meta = []
    for name, text, args in cases:
        naf = O.ref_ennaf(text, args)
        open(os.path.join(HERE, "naf", name + ".naf"), "wb").write(naf)
        if len(text) <= 64 * 1024:
            open(os.path.join(HERE, "naf", name + ".in"), "wb").write(text)
        h = O.parse_naf(naf)
        entry = {"name": name, "ennaf_args": args, "input_sha256": sha(text), "input_len": len(text),
                 "naf_len": len(naf), "outputs": {}}
        for m, margs in modes.items():
            if m == "fastq" and not (h.flags & 1):
                continue
            if m == "4bit" and h.seq_type >= 2:
                continue
            out = O.ref_unnaf(naf, margs)
            entry["outputs"][m] = {"len": len(out), "sha256": sha(out)}
        entry["frame_info"] = {}
        for i, sec in enumerate(("ids", "comments", "lengths", "mask", "seq", "qual")):
            if h.payload_off[i] is not None:
                fi = O.zstd_frame_info(h.frame(naf, i))
                entry["frame_info"][sec] = {"blocks": fi.n_blocks, "raw": fi.n_raw, "rle": fi.n_rle, "comp": fi.n_compressed,
                                            "lit": [fi.lit_raw, fi.lit_rle, fi.lit_huf, fi.lit_treeless],
                                            "nseq": fi.n_sequences, "modes": [list(r) for r in fi.mode_count],
                                            "wlog": fi.window_log, "max_off": fi.max_offset}
        meta.append(entry)
        print(name, len(text), "->", len(naf), entry["frame_info"].get("seq"))
    json.dump(meta, open(os.path.join(HERE, "naf_cases.json"), "w"), indent=1)


def zstd_cases():
    import ctypes
    z = ctypes.CDLL("/opt/conda/lib/libzstd.so.1")
    z.ZSTD_compress.restype = ctypes.c_size_t
    z.ZSTD_compress.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_int]
    z.ZSTD_compressBound.restype = ctypes.c_size_t
    z.ZSTD_compressBound.argtypes = [ctypes.c_size_t]

    def lib_compress(d, lvl):
        cap = z.ZSTD_compressBound(len(d))
        buf = ctypes.create_string_buffer(cap)
        n = z.ZSTD_compress(buf, cap, d, len(d), lvl)
        return buf.raw[:n]

    def cli(d, *args):
        p = subprocess.run([ZSTD_CLI, "-c", "-q", *args], input=d, stdout=subprocess.PIPE, check=True)
        return p.stdout

    os.makedirs(os.path.join(HERE, "zstd"), exist_ok=True)
    rng = np.random.Generator(np.random.PCG64(77))
    packed_syms = np.array([0x88, 0x84, 0x82, 0x81, 0x48, 0x44, 0x42, 0x41, 0x28, 0x24, 0x22, 0x21, 0x18, 0x14, 0x12, 0x11], dtype=np.uint8)
    packed = packed_syms[rng.integers(0, 16, 400001)].tobytes()
    ids = b"".join(b"read%d len=%d\x00" % (i, 100 + i % 50) for i in range(20000))
    qual = rng.integers(33, 74, 300000, dtype=np.uint8).tobytes()
    lens = np.full(50000, 150, dtype="<u4").tobytes()
    p256 = np.array([2.0 ** -(i % 13 + 1) for i in range(256)])
    skew = rng.choice(np.arange(256, dtype=np.uint8), 300000, p=p256 / p256.sum()).tobytes()
    rep = repeat_genome(seed=5, unit=30000, copies=20)
    data = {"packed": packed, "ids": ids, "qual": qual, "lens": lens, "skew": skew, "rep": rep}
    meta = []

    def add(name, frame, content, note):
        open(os.path.join(HERE, "zstd", name + ".zst"), "wb").write(frame)
        info = None
        try:
            fi = O.zstd_frame_info(frame)
            info = {"blocks": fi.n_blocks, "raw": fi.n_raw, "rle": fi.n_rle, "comp": fi.n_compressed,
                    "lit": [fi.lit_raw, fi.lit_rle, fi.lit_huf, fi.lit_treeless], "nseq": fi.n_sequences,
                    "modes": [list(r) for r in fi.mode_count], "wlog": fi.window_log, "single": fi.single_segment,
                    "checksum": fi.has_checksum, "fcs": fi.has_fcs, "max_off": fi.max_offset}
        except ValueError:
            pass
        meta.append({"name": name, "len": len(content), "sha256": sha(content), "zst_len": len(frame), "note": note, "first_frame": info})
        print(name, len(content), "->", len(frame), info and info["lit"], info and info["modes"])

    for dn, d in data.items():
        for lvl in (1, 3, 9, 19):
            if dn in ("packed", "qual", "skew") and lvl in (3, 9):
                continue
            add("%s_l%d" % (dn, lvl), lib_compress(d, lvl), d, "ZSTD_compress level %d (single segment, content size)" % lvl)
    add("empty", lib_compress(b"", 1), b"", "empty content")
    add("one", lib_compress(b"A", 1), b"A", "1 byte")
    add("rle300k", lib_compress(b"\x00" * 300000, 1), b"\x00" * 300000, "RLE blocks")
    add("neg5_ids", lib_compress(ids, -5), ids, "negative level")
    add("cli_mt_packed", cli(packed, "-1", "-T4", "-B1048576", "--zstd=ovlog=1", "--no-check"), packed, "one frame of independent jobs (MT)")
    add("cli_check_ids", cli(ids, "-3", "--check"), ids, "windowed frame with XXH64 checksum")
    add("cli_long27_rep", cli(repeat_genome(seed=11, unit=300000, copies=8), "-3", "--long=27", "--no-check"),
        repeat_genome(seed=11, unit=300000, copies=8), "long-distance matching, windowLog 27")
    add("cli_stream_qual", cli(qual, "-1", "--no-check", "--no-content-size") if False else cli(qual, "-1", "--no-check"), qual, "CLI streaming frame")
    two = lib_compress(ids[:50000], 3) + lib_compress(ids[50000:], 1)
    add("two_frames", two, ids, "two concatenated frames (allowed for one-shot sections, SURVEY R1)")
    skippable = b"\x50\x2a\x4d\x18" + (7).to_bytes(4, "little") + b"ignored" + lib_compress(lens, 1)
    add("skippable_then_frame", skippable, lens, "skippable frame first")
    json.dump(meta, open(os.path.join(HERE, "zstd_cases.json"), "w"), indent=1)


if __name__ == "__main__":
    assert O.have_ref(), "build oracle/_ref first (make -C oracle ref)"
    copy_ref_tests()
    naf_cases()
    zstd_cases()
    tot = 0
    for dp, _, fs in os.walk(HERE):
        tot += sum(os.path.getsize(os.path.join(dp, f)) for f in fs)
    print("golden bytes:", tot)
