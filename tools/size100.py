#!/usr/bin/env python3
"""The metric's own size on ONE GPU (BASELINE.json: "on 100 GB synthetic FASTA"): tools/size100.py [bytes] [steps]
100 GB of synthetic-ACGT FASTA encoded by one naf_gpu_ennaf call, the archive decoded by one naf_gpu_unnaf call per step -- text, archive
and decoded text resident in HBM (225 GB of the 288).  Checks: the decoded text against the input chunk by chunk, and the position-
weighted checksum bench.py uses for its sharded runs.  Prints one JSON line (kept as profiles/rNN_size100_line.json)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from naf_amd import capi, synth

size = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(100e9)
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ctx = capi.Context(0)
text = synth.fasta_acgt_device(size, n_records=100, width=80, seed=2024, device="cuda")
n = int(text.numel())
naf_buf = torch.empty(int(n * 0.27) + (1 << 20), dtype=torch.uint8, device="cuda")
enc = []
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    d_naf, rep = ctx.ennaf(text, out=naf_buf)
    torch.cuda.synchronize(); enc.append(time.perf_counter() - t0)
n_naf = int(d_naf.numel())
free0 = torch.cuda.mem_get_info()[0]
ctx.release_scratch()                                   # the encoder's arena (the packed stream of 100 GB of text among it) before the text is made a second time
wsum = bench.weighted_sum(text, 0)
out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
r = ctx.unnaf(d_naf, capi.OUT_FASTA, out=out)
torch.cuda.synchronize()
same = int(r.numel()) == n
step = 1 << 30
for a in range(0, n, step):
    same = same and bool(torch.equal(r[a:a + step], text[a:a + step]))
wsum_back = bench.weighted_sum(r, 0)
for _ in range(5):
    ctx.unnaf(d_naf, capi.OUT_FASTA, out=out)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    ctx.unnaf(d_naf, capi.OUT_FASTA, out=out)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
ctx.set_timing(True); ctx.unnaf(d_naf, capi.OUT_FASTA, out=out)
kt = sorted(ctx.get_timing(), key=lambda x: -x[1])[:8]; streams = ctx.get_timing_streams(); ctx.set_timing(False)
emit = dict((nm, ms) for nm, ms, k in kt).get("unnaf_emit_flat", 0.0)
line = {"metric": "unnaf GB/s (uncompressed bases out) on synthetic FASTA", "value": round(n / dt / 1e9, 3), "unit": "GB/s", "n_gpus": 1, "steps": steps, "warmup": 5,
        "ms_per_step": round(dt * 1e3, 3), "text_bytes": n, "naf_bytes": n_naf, "naf_ratio": round(n_naf / n, 4), "records": 100, "bases": int(rep.n_bases),
        "roundtrip_bit_exact": bool(same), "weighted_sum_equal": bool(wsum == wsum_back), "weighted_sum": int(wsum),
        "ennaf_value": round(n / sorted(enc)[len(enc) // 2] / 1e9, 3), "ennaf_ms": [round(x * 1e3, 2) for x in enc],
        "roofline": {"kernel": "unnaf_emit_flat", "kernel_ms_per_step": round(emit, 3), "algorithmic_bytes_per_launch": int(rep.section_comp[4]) + n,
                     "frac": round((int(rep.section_comp[4]) + n) / (emit * 1e-3) / bench.HBM_PEAK, 4) if emit else None,
                     "path_frac": round((n + n_naf) / dt / bench.HBM_PEAK, 4), "stream_kernel_ms": [round(x, 3) for x in streams]},
        "kernels_ms": {nm: round(ms, 3) for nm, ms, k in kt}, "hbm_free_after_encode_gb": round(free0 / 1e9, 1), "box": bench.box_id()}
print(json.dumps(line), flush=True)
