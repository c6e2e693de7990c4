#!/usr/bin/env python3
"""One of bench.py's side workloads alone, with per-call times and the kernel lists: tools/perf_side.py fastq|realistic|softmasked [bytes]
(run on the GPU box from the repo root; environment switches of the library apply, e.g. NAF_GPU_PREFER_FLAT=0)."""
import json
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
    text = synth.fastq_reads_device(size, seed=7, device="cuda"); mode = capi.OUT_FASTQ
elif which == "realistic":
    text = synth.realistic_genome_device(size, device="cuda"); mode = capi.OUT_FASTA
elif which == "uniform":
    text = synth.fasta_acgt_device(size, n_records=100, width=80, seed=2024, device="cuda"); mode = capi.OUT_FASTA      # the bench line's headline text
else:
    text = synth.softmask_device(synth.fasta_acgt_device(size, n_records=24, width=60, seed=7, device="cuda")); mode = capi.OUT_FASTA
n = text.numel()
ctx.reserve(int(n * 3.0) + (1 << 30))
buf = torch.empty(int(ctx.L.naf_gpu_ennaf_bound(n)), dtype=torch.uint8, device="cuda")
for it in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d_naf, rep = ctx.ennaf(text, out=buf)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("ennaf call %d: %.2f ms  %.1f GB/s  (naf %d B, %.4f)" % (it, dt * 1e3, n / dt / 1e9, d_naf.numel(), d_naf.numel() / n), flush=True)
ctx.set_timing(True); ctx.ennaf(text, out=buf)
for nm, ms, k in sorted(ctx.get_timing(), key=lambda x: -x[1])[:16]:
    print("  ENC %-26s %8.3f ms x%d" % (nm, ms, k))
ctx.set_timing(False)
# a checksum of the archive: two builds / switches that claim the same archive can be compared from their logs (behind the instrumented
# call: tools/trace_step.sh cuts the last timed call out of the trace between two k_sniff launches)
w = (torch.arange(d_naf.numel(), device="cuda", dtype=torch.int64) % 65521) + 1
print("archive checksum: %d" % int((d_naf.to(torch.int64) * w).sum().item()))
del w
d_naf = d_naf.clone()
out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
for it in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ctx.unnaf(d_naf, mode, out=out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("unnaf call %d: %.2f ms  %.1f GB/s" % (it, dt * 1e3, n / dt / 1e9), flush=True)
ctx.set_timing(True); ctx.unnaf(d_naf, mode, out=out)
for nm, ms, k in sorted(ctx.get_timing(), key=lambda x: -x[1])[:18]:
    print("  DEC %-26s %8.3f ms x%d" % (nm, ms, k))
ctx.set_timing(False)
same = bool(((r == text) | ((r ^ 32) == text)).all()) if which == "fastq" else bool(torch.equal(r, text))
print("round trip ok:", same)
