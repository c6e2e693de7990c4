#!/usr/bin/env python3
"""The decode of a REFERENCE-made archive (oracle/_ref/ennaf on the host with its default level, then this build's unnaf on the GPU):
tools/perf_refarc.py [bytes] [uniform|realistic|repeats|fastq] [ennaf flags ...]  -- per-call times and the kernel list; NAF_GPU_TRACE=1
shows how many blocks were decoded."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from naf_amd import capi, synth

size = int(float(sys.argv[1])) if len(sys.argv) > 1 else int(4e9)
which = sys.argv[2] if len(sys.argv) > 2 else "uniform"
flags = sys.argv[3:]
mode = capi.OUT_FASTQ if which == "fastq" else capi.OUT_FASTA
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ctx = capi.Context(0)
text = (synth.fasta_acgt_device(size, n_records=100, width=80, seed=2024, device="cuda") if which == "uniform" else
        synth.realistic_genome_device(size, device="cuda") if which == "realistic" else
        synth.repeat_genome_device(size, device="cuda") if which == "repeats" else synth.fastq_reads_device(size, seed=7, device="cuda"))
n = text.numel()
d = "/dev/shm/refarc_%d" % os.getpid(); os.makedirs(d, exist_ok=True)
try:
    text.cpu().numpy().tofile(d + "/t.fa")
    t0 = time.perf_counter()
    subprocess.check_call([root + "/oracle/_ref/ennaf", *flags, d + "/t.fa", "-o", d + "/t.naf"], env=dict(os.environ, TMPDIR=d))
    print("reference ennaf: %.1f s, archive %d B" % (time.perf_counter() - t0, os.path.getsize(d + "/t.naf")))
    t0 = time.perf_counter(); subprocess.check_call([root + "/oracle/_ref/unnaf", d + "/t.naf", "-o", d + "/t.out"]); tr = time.perf_counter() - t0
    print("reference unnaf: %.2f s = %.2f GB/s" % (tr, n / tr / 1e9))
    want = torch.from_numpy(np.fromfile(d + "/t.out", dtype=np.uint8)).to("cuda")
    naf = torch.from_numpy(np.fromfile(d + "/t.naf", dtype=np.uint8)).to("cuda")
finally:
    subprocess.call(["rm", "-rf", d])
ctx.reserve(int(n * 1.7) + (2 << 30))
out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
for it in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    r = ctx.unnaf(naf, mode, out=out)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("unnaf call %d: %.2f ms  %.1f GB/s" % (it, dt * 1e3, n / dt / 1e9), flush=True)
print("bit-exact with the reference's output:", bool(torch.equal(r, want)))
ctx.set_timing(True); ctx.unnaf(naf, mode, out=out)
for nm, ms, k in sorted(ctx.get_timing(), key=lambda x: -x[1])[:22]:
    print("   %-28s %8.3f ms x%d" % (nm, ms, k))
print("   streams:", ["%.3f" % x for x in ctx.get_timing_streams()])
ctx.set_timing(False)
if os.environ.get("RANGE8"):                     # an eighth of it by byte range (what a rank of an 8-GPU job decodes), the fourth eighth
    b = n * 3 // 8 // 4096 * 4096; e = b + n // 8
    o8 = torch.empty(e - b + 64, dtype=torch.uint8, device="cuda")
    for it in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r8 = ctx.unnaf_range(naf, b, e, mode, out=o8)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("range call %d: %.3f ms" % (it, dt * 1e3), flush=True)
    print("bit-exact:", bool(torch.equal(r8, want[b:e])))
    ctx.set_timing(True); ctx.unnaf_range(naf, b, e, mode, out=o8)
    for nm, ms, k in sorted(ctx.get_timing(), key=lambda x: -x[1])[:22]:
        print("   %-28s %8.3f ms x%d" % (nm, ms, k))
    print("   streams:", ["%.3f" % x for x in ctx.get_timing_streams()])
    ctx.set_timing(False)
