#!/usr/bin/env python3
"""Two values of one of the library's switches, called turn by turn on the same data:
tools/perf_ab.py NAF_GPU_POLL 0 1 [uniform|fastq|realistic|softmasked] [bytes] [same]
Two contexts of one process, each made under its value (a switch a context reads when it is made) -- the context made first tends to be
the faster one by a few per cent, so run both orders -- or, with `same`, ONE context and the switch set before each call (a switch the
library reads per call: most of them).  Prints the median / min of ten calls each for ennaf and unnaf."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from naf_amd import capi, synth

name, va, vb = sys.argv[1], sys.argv[2], sys.argv[3]
which = sys.argv[4] if len(sys.argv) > 4 else "uniform"
size = int(float(sys.argv[5])) if len(sys.argv) > 5 and sys.argv[5] != "same" else int(10e9 if which == "uniform" else 4e9)
same = "same" in sys.argv[5:]
mode = capi.OUT_FASTA
if which == "fastq":
    text = synth.fastq_reads_device(size, seed=7, device="cuda"); mode = capi.OUT_FASTQ
elif which == "realistic":
    text = synth.realistic_genome_device(size, device="cuda")
elif which == "uniform":
    text = synth.fasta_acgt_device(size, n_records=100, width=80, seed=2024, device="cuda")
else:
    text = synth.softmask_device(synth.fasta_acgt_device(size, n_records=24, width=60, seed=7, device="cuda"))
n = text.numel()
ctxs = []
for v in (va, vb):
    os.environ[name] = v
    if same and ctxs: ctxs.append(ctxs[0]); break
    c = capi.Context(0); c.reserve(int(n * 3.0) + (1 << 30)); ctxs.append(c)
buf = torch.empty(int(ctxs[0].L.naf_gpu_ennaf_bound(n)), dtype=torch.uint8, device="cuda")
out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
te = ([], []); td = ([], [])
d_naf = None
for it in range(13):
    for k, c in enumerate(ctxs):
        os.environ[name] = (va, vb)[k]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        a, rep = c.ennaf(text, out=buf)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if it >= 3: te[k].append(dt * 1e3)
        if d_naf is None: d_naf = a.clone()
for it in range(13):
    for k, c in enumerate(ctxs):
        os.environ[name] = (va, vb)[k]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = c.unnaf(d_naf, mode, out=out)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if it >= 3: td[k].append(dt * 1e3)
for k, v in enumerate((va, vb)):
    print("%s=%s  %s %d B: ennaf median %.3f min %.3f ms | unnaf median %.3f min %.3f ms" % (name, v, which, n, statistics.median(te[k]), min(te[k]), statistics.median(td[k]), min(td[k])))
