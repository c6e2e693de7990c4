#!/usr/bin/env python3
"""The split kernels of a FASTQ encode alone: tools/perf_fqsplit.py [bytes]  (NAF_GPU_LIB picks the build; a build whose split writes
nonsense -- an ablation -- may fail behind the split: the kernel times are printed all the same)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from naf_amd import capi, synth

size = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(4e9)
ctx = capi.Context(0)
text = synth.fastq_reads_device(size, seed=7, device="cuda")
n = text.numel()
ctx.reserve(int(n * 3.0) + (1 << 30))
buf = torch.empty(int(ctx.L.naf_gpu_ennaf_bound(n)), dtype=torch.uint8, device="cuda")
for it in range(3):
    ctx.set_timing(True)
    try:
        ctx.ennaf(text, out=buf)
    except Exception as e:
        print("ennaf failed:", str(e)[:80])
    kt = ctx.get_timing()
    ctx.set_timing(False)
print(os.path.basename(os.environ.get("NAF_GPU_LIB", "default")), " ".join("%s %.3f" % (nm.replace("ennaf_", ""), ms) for nm, ms, k in kt if nm.startswith("ennaf_") and ms > 0.05))
