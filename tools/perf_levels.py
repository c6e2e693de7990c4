#!/usr/bin/env python3
"""tools/perf_levels.py <bytes> [ennaf flags ...]: a repeat-rich genome of that size, the REFERENCE's archive of it with the given flags
(default: --level 3 --long 27), decoded here: bit-exact or the error, the time of a call and its longest kernels."""
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from naf_amd import capi, synth

size = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(1e9)
flags = sys.argv[2:] or ["--level", "3", "--long", "27"]
ctx = capi.Context(0)
text = synth.repeat_genome_device(size, device="cuda")
n = int(text.numel())
shm = "/dev/shm/naf_lv_%d" % os.getpid(); os.makedirs(shm, exist_ok=True)
try:
    text.cpu().numpy().tofile(shm + "/t.fa")
    t0 = time.perf_counter(); subprocess.check_call([bench.REF_E, *flags, shm + "/t.fa", "-o", shm + "/t.naf"], env=dict(os.environ, TMPDIR=shm)); tr = time.perf_counter() - t0
    naf = torch.from_numpy(np.fromfile(shm + "/t.naf", dtype=np.uint8)).cuda()
    print("text", n, "reference archive", int(naf.numel()), "reference ennaf %.2f s" % tr, flush=True)
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    try:
        t0 = time.perf_counter(); r = ctx.unnaf(naf, 0, out=buf); torch.cuda.synchronize(); print("first call %.1f ms" % ((time.perf_counter() - t0) * 1e3))
        print("bit exact", bool(torch.equal(r, text)))
        ts = bench.timed_calls(lambda: ctx.unnaf(naf, 0, out=buf), 3, warm=1)
        print("unnaf %.2f ms = %.1f GB/s" % (bench.median(ts) * 1e3, n / bench.median(ts) / 1e9))
        top, _all, streams = bench.instrumented(ctx, lambda: ctx.unnaf(naf, 0, out=buf), top=8)
        print([(k, round(ms, 3), c) for k, ms, c in top], [round(x, 2) for x in streams])
    except capi.NafGpuError as ex:
        print("ERROR", ex)
    for lv, kw in (("1", {}), ("19", {"level": 19}), ("3 long 27", {"level": 3, "long_log": 27})):
        res = [None]
        def f():
            res[0] = ctx.ennaf(text, **kw)
        ts = bench.timed_calls(f, 2, warm=1)
        mine = res[0][0]
        back = ctx.unnaf(mine, 0, out=buf); torch.cuda.synchronize()
        print("gpu ennaf level", lv, "archive", int(mine.numel()), "%.1f ms = %.2f GB/s" % (bench.median(ts) * 1e3, n / bench.median(ts) / 1e9), "own round trip", bool(torch.equal(back, text)), flush=True)
finally:
    subprocess.call(["rm", "-rf", shm])
