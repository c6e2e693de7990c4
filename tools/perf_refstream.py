#!/usr/bin/env python3
"""Every section of a REFERENCE-made archive decoded alone (naf_gpu_zstd_decompress on the section's frame) with its longest kernels:
tools/perf_refstream.py [bytes] [fastq|realistic|repeats|uniform] [ennaf flags ...]"""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from naf_amd import capi, synth
size = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(2e9)
which = sys.argv[2] if len(sys.argv) > 2 else "fastq"
flags = sys.argv[3:]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ctx = capi.Context(0)
text = (synth.fasta_acgt_device(size, n_records=100, width=80, seed=2024, device="cuda") if which == "uniform" else
        synth.realistic_genome_device(size, device="cuda") if which == "realistic" else
        synth.repeat_genome_device(size, device="cuda") if which == "repeats" else synth.fastq_reads_device(size, seed=7, device="cuda"))
d = "/dev/shm/refs_%d" % os.getpid(); os.makedirs(d, exist_ok=True)
try:
    text.cpu().numpy().tofile(d + "/t.fa")
    subprocess.check_call([root + "/oracle/_ref/ennaf", *flags, d + "/t.fa", "-o", d + "/t.naf"], env=dict(os.environ, TMPDIR=d), stderr=subprocess.DEVNULL)
    naf = torch.from_numpy(np.fromfile(d + "/t.naf", dtype=np.uint8)).to("cuda")
finally:
    subprocess.call(["rm", "-rf", d])
del text
h = ctx.parse_header(naf)
names = ["ids", "names", "lengths", "mask", "sequence", "quality"]
for i in range(6):
    if not h.comp_size[i]:
        continue
    frame = naf[h.payload_off[i]:h.payload_off[i] + h.comp_size[i]]
    cap = int(h.orig_size[i]) if i != 4 else (int(h.orig_size[i]) + 1) // 2
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = ctx.zstd_decompress(frame, cap + 64, has_magic=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    ctx.set_timing(True); ctx.zstd_decompress(frame, cap + 64, has_magic=False)
    tm = sorted(ctx.get_timing(), key=lambda x: -x[1])[:6]
    ctx.set_timing(False)
    print("%-9s %11d -> %11d B  %8.3f ms   %s" % (names[i], h.comp_size[i], out.numel(), dt * 1e3, "  ".join("%s %.3f" % (n, ms) for n, ms, k in tm)), flush=True)
    del out
