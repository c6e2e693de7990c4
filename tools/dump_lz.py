import os, sys, subprocess
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/naf_amd") else ".")
import numpy as np, torch
from naf_amd import capi, synth
from oracle import oracle as O
os.environ["NAF_GPU_TRACE"] = "1"
ctx = capi.Context(0)
text = synth.fastq_reads(200_000, 150, seed=3)
naf = O.ref_ennaf(text, ("--fastq",))
d = ctx.to_device(naf)
for coll in ("1", "0"):
    os.environ["NAF_GPU_EXEC_COLLAPSE"] = coll
    print("=== collapse", coll, flush=True)
    r = ctx.unnaf(d, capi.OUT_FASTQ)
    torch.cuda.synchronize()
