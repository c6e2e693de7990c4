#!/usr/bin/env python3
"""The kernels of the shard protocol's two calls on one device: tools/perf_shard.py [bytes] [shards]  (shards contexts on device 0, one
after the other; prints the kernel lists of shard 0's begin and finish calls beside the one-call encode of the same text)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from naf_amd import capi, synth, shard

size = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(10e9)
ns = int(sys.argv[2]) if len(sys.argv) > 2 else 1
text = synth.fasta_acgt_device(size, n_records=100, width=80, seed=2024, device="cuda")
n = text.numel()
ctxs = [capi.Context(0) for _ in range(ns)]
for c in ctxs: c.reserve(int(n * 3.0 / ns) + (1 << 30))
opts = shard.make_opts()
c0 = ctxs[0]
fmt, p0 = c0.ennaf_sniff(text, opts.format)
cuts = shard.cuts_local(c0, text, fmt, p0, ns) if ns > 1 else [p0, n]
slices = [text[cuts[k]:cuts[k + 1]] for k in range(ns)]
for it in range(4):
    tm = it == 3
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if tm: c0.set_timing(True)
    infos = [ctxs[k].ennaf_shard_begin(slices[k], opts, fmt, k, ns) for k in range(ns)]
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if tm:
        kb = c0.get_timing(); c0.set_timing(False); c0.set_timing(True)
    outs = [ctxs[k].ennaf_shard_finish(opts, infos, slices[k].numel()) for k in range(ns)]
    torch.cuda.synchronize(); t2 = time.perf_counter()
    if tm:
        kf = c0.get_timing(); c0.set_timing(False)
    print("begin %.3f ms  finish %.3f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3))
for nm, ms, k in sorted(kb, key=lambda x: -x[1])[:14]: print("  BEGIN  %-28s %8.3f ms x%d" % (nm, ms, k))
for nm, ms, k in sorted(kf, key=lambda x: -x[1])[:14]: print("  FINISH %-28s %8.3f ms x%d" % (nm, ms, k))
c0.set_timing(True); c0.ennaf(text); kt = c0.get_timing(); c0.set_timing(False)
for nm, ms, k in sorted(kt, key=lambda x: -x[1])[:14]: print("  ONE    %-28s %8.3f ms x%d" % (nm, ms, k))
