#!/bin/bash
# Kernel timelines of the decode of a REFERENCE-made archive of uniform bases: the last whole call and the last call for an eighth of
# the text by byte range (what a rank of an 8-GPU job runs).   tools/trace_refarc.sh [bytes]   (GPU box, repo root)
size=${1:-10e9}
mkdir -p gpurun_out/trace_refarc; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_refarc/raw -- python - $size > $GRAFT_REPO_ROOT/gpurun_out/trace_refarc/run.log 2>&1 <<'PY'
import os, subprocess, sys, time
root = os.environ["GRAFT_REPO_ROOT"]; sys.path.insert(0, root)
import numpy as np, torch
from naf_amd import capi, synth
size = int(float(sys.argv[1]))
ctx = capi.Context(0)
text = synth.fasta_acgt_device(size, n_records=100, width=80, seed=2024, device="cuda")
n = text.numel()
d = "/dev/shm/trr_%d" % os.getpid(); os.makedirs(d, exist_ok=True)
try:
    text.cpu().numpy().tofile(d + "/t.fa")
    subprocess.check_call([root + "/oracle/_ref/ennaf", d + "/t.fa", "-o", d + "/t.naf"], env=dict(os.environ, TMPDIR=d))
    naf = torch.from_numpy(np.fromfile(d + "/t.naf", dtype=np.uint8)).to("cuda")
finally:
    subprocess.call(["rm", "-rf", d])
ctx.reserve(int(n * 1.7) + (2 << 30))
out = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
for it in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = ctx.unnaf(naf, capi.OUT_FASTA, out=out); torch.cuda.synchronize()
    print("whole %.3f ms" % ((time.perf_counter() - t0) * 1e3))
print("bit-exact", bool(torch.equal(r, text)))
b = n * 3 // 8 // 4096 * 4096; e = b + n // 8
for it in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = ctx.unnaf_range(naf, b, e, capi.OUT_FASTA, out=out); torch.cuda.synchronize()
    print("eighth %.3f ms" % ((time.perf_counter() - t0) * 1e3))
print("bit-exact", bool(torch.equal(r, text[b:e])))
PY
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace_refarc/raw/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(f))), key=lambda x: x[0])
cands = [i for i, r in enumerate(rows) if r[2].startswith("k_parse_container")]
def dump(lo, hi, name):
    t0 = rows[lo][0]
    with open("gpurun_out/trace_refarc/%s.txt" % name, "w") as o:
        for s, e, n, q in rows[lo:hi]:
            o.write("%9.1f us  +%8.1f us  q%-3s %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
        o.write("total %.1f us, %d launches\n" % ((max(r[1] for r in rows[lo:hi]) - t0) / 1e3, hi - lo))
dump(cands[4], cands[5], "timeline_whole")          # the fifth whole call
dump(cands[-1], len(rows), "timeline_eighth")       # the last range call
PY
rm -rf gpurun_out/trace_refarc/raw
cat gpurun_out/trace_refarc/run.log | tail -14
