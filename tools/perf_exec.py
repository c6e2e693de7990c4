#!/usr/bin/env python3
"""tools/perf_exec.py <bytes> : the reference's --long 27 archive of a repeat-rich genome decoded with each sequence executor
(NAF_GPU_EXEC = dataflow (default) with units of 4 / 8 / 16 words, batch, serial) -- DESIGN.md 4.30."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, bench
from naf_amd import capi, synth
size = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(2e8)
ctx = capi.Context(0)
text = synth.repeat_genome_device(size, device="cuda"); n = int(text.numel())
shm = "/dev/shm/naf_px_%d" % os.getpid(); os.makedirs(shm, exist_ok=True)
try:
    text.cpu().numpy().tofile(shm + "/t.fa")
    subprocess.check_call([bench.REF_E, "--level", "3", "--long", "27", shm + "/t.fa", "-o", shm + "/t.naf"], env=dict(os.environ, TMPDIR=shm))
    t0 = time.perf_counter(); subprocess.check_call([bench.REF_U, shm + "/t.naf", "-o", shm + "/t.out"]); tr = time.perf_counter() - t0
    print("text %d B; the reference decodes its archive in %.2f s = %.2f GB/s" % (n, tr, n / tr / 1e9), flush=True)
    naf = torch.from_numpy(np.fromfile(shm + "/t.naf", dtype=np.uint8)).cuda()
    buf = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
    modes = sys.argv[2:] or ["dataflow:16", "dataflow:8", "dataflow:4", "batch", "serial"]
    for mode in modes:
        how, _, unit = mode.partition(":")
        os.environ["NAF_GPU_EXEC"] = how
        if unit: os.environ["NAF_GPU_EXEC_UNIT"] = unit
        try:
            buf.zero_()
            r = ctx.unnaf(naf, 0, out=buf); torch.cuda.synchronize(); ok = bool(torch.equal(r, text))
            ts = bench.timed_calls(lambda: ctx.unnaf(naf, 0, out=buf), 3, warm=1)
            top, _a, streams = bench.instrumented(ctx, lambda: ctx.unnaf(naf, 0, out=buf), top=5)
            print(mode, "bit exact", ok, "%.2f ms = %.1f GB/s" % (bench.median(ts) * 1e3, n / bench.median(ts) / 1e9), [(k, round(ms, 2)) for k, ms, c in top], flush=True)
        except capi.NafGpuError as ex:
            print(mode, "ERROR", ex, flush=True)
finally:
    subprocess.call(["rm", "-rf", shm])
