#!/bin/bash
# Kernel timeline of the LAST decode call of tools/perf_side.py (rocprofv3 --kernel-trace): every launch with its start relative to
# the first launch of that call and its duration.   tools/trace_step.sh <workload> <bytes> [tag]   (GPU box, repo root)
which=${1:-realistic}; size=${2:-4e9}; tag=${3:-$which}
mkdir -p gpurun_out/trace_$tag; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_$tag/raw -- python $GRAFT_REPO_ROOT/tools/perf_side.py $which $size > $GRAFT_REPO_ROOT/gpurun_out/trace_$tag/run.log 2>&1
cd $GRAFT_REPO_ROOT
python - "$tag" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob("gpurun_out/trace_%s/raw/**/*kernel_trace.csv" % tag, recursive=True)[0]
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Stream_Id", r.get("Queue_Id", "?"))) for r in csv.DictReader(open(f))), key=lambda x: x[0])
# the last decode call: from the last k_parse_container on
idx = max(i for i, r in enumerate(rows) if r[2].startswith("k_parse_container"))
# the instrumented call is the last one; take the one before it (the last timed call) when there are two
cands = [i for i, r in enumerate(rows) if r[2].startswith("k_parse_container")]
if len(cands) >= 2: lo, hi = cands[-2], cands[-1]
else: lo, hi = cands[-1], len(rows)
t0 = rows[lo][0]
with open("gpurun_out/trace_%s/timeline.txt" % tag, "w") as o:
    for s, e, n, q in rows[lo:hi]:
        o.write("%9.1f us  +%8.1f us  q%-3s %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
    o.write("total %.1f us, %d launches\n" % ((max(r[1] for r in rows[lo:hi]) - t0) / 1e3, hi - lo))
# the last timed encode call: from the fifth k_sniff (perf_side.py makes five timed calls and an instrumented one) to the sixth
sn = [i for i, r in enumerate(rows) if r[2].startswith("k_sniff")]
if len(sn) >= 2:
    lo, hi = sn[-2], sn[-1]
    t0 = rows[lo][0]
    with open("gpurun_out/trace_%s/timeline_ennaf.txt" % tag, "w") as o:
        for s, e, n, q in rows[lo:hi]:
            o.write("%9.1f us  +%8.1f us  q%-3s %s\n" % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
        o.write("total %.1f us, %d launches\n" % ((max(r[1] for r in rows[lo:hi]) - t0) / 1e3, hi - lo))
PY
rm -rf gpurun_out/trace_$tag/raw
