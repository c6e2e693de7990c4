#!/usr/bin/env python3
"""The small sections (ids, names, lengths, mask) of the headline archive one by one through the one-lane small-frame decoder:
tools/perf_small.py [records]  -- prints each frame's sizes and the kernel time of its decode."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from naf_amd import capi, synth

nrec = int(sys.argv[1]) if len(sys.argv) > 1 else 100
ctx = capi.Context(0)
text = synth.fasta_acgt_device(int(2e8), n_records=nrec, width=80, seed=2024, device="cuda")
d_naf, rep = ctx.ennaf(text)
d_naf = d_naf.clone()
h = ctx.parse_header(d_naf)
print("first header:", bytes(text[:60].cpu().numpy()))
for k, nm in enumerate(("ids", "names", "lengths", "mask", "seq", "qual")):
    o, cs, off = h.orig_size[k], h.comp_size[k], h.payload_off[k]
    if not cs or cs > 60000: print("%-8s orig %d comp %d" % (nm, o, cs)); continue
    fr = d_naf[off:off + cs].clone()
    for it in range(3):
        ctx.set_timing(True); out = ctx.zstd_decompress(fr, int(o), has_magic=False); kt = ctx.get_timing(); ctx.set_timing(False)
    print("%-8s orig %d comp %d:" % (nm, o, cs), " ".join("%s %.1f us x%d" % (a, b * 1e3, c) for a, b, c in kt))
