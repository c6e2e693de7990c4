#!/usr/bin/env python3
"""Every section of an archive COMPRESSED alone (its stream taken back out of the archive, then naf_gpu_zstd_compress on it, with the match
finder for the streams ennaf gives it to): what a stream's chain of encoder kernels costs with the device to itself -- the other half of
tools/perf_stream.py.   tools/perf_encstream.py fastq|realistic|softmasked|uniform [bytes]   (GPU box, repo root)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from naf_amd import capi, synth

which = sys.argv[1] if len(sys.argv) > 1 else "fastq"
size = int(float(sys.argv[2])) if len(sys.argv) > 2 else int(4e9)
ctx = capi.Context(0)
if which == "fastq":
    text = synth.fastq_reads_device(size, seed=7, device="cuda")
elif which == "realistic":
    text = synth.realistic_genome_device(size, device="cuda")
elif which == "uniform":
    text = synth.fasta_acgt_device(size, n_records=24, width=80, seed=5, device="cuda")
else:
    text = synth.softmask_device(synth.fasta_acgt_device(size, n_records=24, width=60, seed=7, device="cuda"))
ctx.reserve(int(text.numel() * 3.0) + (1 << 30))
d_naf, rep = ctx.ennaf(text)
d_naf = d_naf.clone()
del text
h = ctx.parse_header(d_naf)
names = ["ids", "names", "lengths", "mask", "sequence", "quality"]
for i in range(6):
    if not h.comp_size[i]:
        continue
    frame = d_naf[h.payload_off[i]:h.payload_off[i] + h.comp_size[i]]
    cap = int(h.orig_size[i]) if i != 4 else (int(h.orig_size[i]) + 1) // 2
    raw = ctx.zstd_decompress(frame, cap + 64, has_magic=False).clone()
    if i < 3: os.environ["NAF_GPU_LZ"] = "all"
    else: os.environ.pop("NAF_GPU_LZ", None)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = ctx.zstd_compress(raw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ctx.set_timing(True); ctx.zstd_compress(raw)
    tm = sorted(ctx.get_timing(), key=lambda x: -x[1])[:7]
    ctx.set_timing(False)
    print("%-9s %11d -> %11d B (in the archive %11d)  %7.3f ms   %s" % (names[i], raw.numel(), out.numel(), h.comp_size[i], dt * 1e3, "  ".join("%s %.3f" % (n, ms) for n, ms, k in tm)), flush=True)
    del out, raw
