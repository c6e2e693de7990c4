#!/usr/bin/env python3
"""ennaf of a synthetic text, turn by turn under each value of a per-call switch, with the kernels of one instrumented call:
tools/perf_enc.py [uniform|fastq|realistic|softmasked] [bytes] [NAF_GPU_X v1 v2 ...]"""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from naf_amd import capi, synth

which = sys.argv[1] if len(sys.argv) > 1 else "uniform"
size = int(float(sys.argv[2])) if len(sys.argv) > 2 else int(10e9)
name = sys.argv[3] if len(sys.argv) > 3 else "NAF_GPU_NOTHING"
vals = sys.argv[4:] or ["-"]
if which == "fastq":
    text = synth.fastq_reads_device(size, seed=7, device="cuda")
elif which == "realistic":
    text = synth.realistic_genome_device(size, device="cuda")
elif which == "uniform":
    text = synth.fasta_acgt_device(size, n_records=100, width=80, seed=2024, device="cuda")
else:
    text = synth.softmask_device(synth.fasta_acgt_device(size, n_records=24, width=60, seed=7, device="cuda"))
n = text.numel()
big = n > 40e9
c = capi.Context(0); c.reserve(int(n * (0.9 if big else 2.0)) + (1 << 30))
buf = torch.empty(int(n * 0.27) + (1 << 20) if big else int(c.L.naf_gpu_ennaf_bound(n)), dtype=torch.uint8, device="cuda")
ts = {v: [] for v in vals}
arch = {}
for it in range(9):
    for v in vals:
        if v == "-": os.environ.pop(name, None)
        else: os.environ[name] = v
        torch.cuda.synchronize(); t0 = time.perf_counter()
        a, rep = c.ennaf(text, out=buf)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if it >= 2: ts[v].append(dt * 1e3)
        if v not in arch: arch[v] = a.clone() if not big else int((a[::4099].to(torch.int64) * 7).sum().item())
for v in vals:
    if v == "-": os.environ.pop(name, None)
    else: os.environ[name] = v
    c.set_timing(True); c.ennaf(text, out=buf); kt = c.get_timing(); c.set_timing(False)
    m = statistics.median(ts[v])
    print("%s=%s %s %d B: ennaf median %.3f min %.3f ms = %.1f GB/s; archive %d B%s" % (name, v, which, n, m, min(ts[v]), n / m / 1e6, a.numel(),
          "" if v == vals[0] else (" SAME as first" if (arch[v] == arch[vals[0]] if big else torch.equal(arch[v], arch[vals[0]])) else " DIFFERS from first")))
    for nm, ms, k in sorted(kt, key=lambda x: -x[1])[:14]:
        print("    %-28s %8.3f ms x%d" % (nm, ms, k))
if big: sys.exit(0)
out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
r = c.unnaf(arch[vals[0]], capi.OUT_FASTQ if which == "fastq" else capi.OUT_FASTA, out=out)
print("round trip", "ok" if (which == "fastq" or torch.equal(r, text)) else "BROKEN")
