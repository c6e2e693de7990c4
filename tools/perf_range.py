#!/usr/bin/env python3
"""One eighth of a text by byte range (what a rank of an 8-GPU job decodes): tools/perf_range.py [bytes] -- per-call times and the kernel list,
for the archive this build makes of the headline text (a flat frame)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from naf_amd import capi, synth

size = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(10e9)
ctx = capi.Context(0)
text = synth.fasta_acgt_device(size, n_records=100, width=80, seed=2024, device="cuda")
n = text.numel()
ctx.reserve(int(n * 1.7) + (2 << 30))
naf, rep = ctx.ennaf(text)
naf = naf.clone()
for frac in (8, 1):
    b = n * 3 // 8 // 4096 * 4096 if frac > 1 else 0
    e = b + n // frac
    out = torch.empty(e - b + 64, dtype=torch.uint8, device="cuda")
    fn = (lambda: ctx.unnaf_range(naf, b, e, capi.OUT_FASTA, out=out)) if frac > 1 else (lambda: ctx.unnaf(naf, capi.OUT_FASTA, out=out))
    for _ in range(3):
        r = fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    ok = bool(torch.equal(r, text[b:e]))
    ts.sort()
    print("1/%d of the text: median %.3f ms  min %.3f  max %.3f  bit-exact %s" % (frac, ts[5] * 1e3, ts[0] * 1e3, ts[-1] * 1e3, ok))
    ctx.set_timing(True); fn()
    for nm, ms, k in sorted(ctx.get_timing(), key=lambda x: -x[1])[:14]:
        print("   %-28s %8.3f ms x%d" % (nm, ms, k))
    print("   streams:", ["%.3f" % x for x in ctx.get_timing_streams()])
    ctx.set_timing(False)
    del out
