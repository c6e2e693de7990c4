#!/bin/bash
# Instruction and wait counters of every kernel of one workload (rocprofv3 --pmc, no tracing domains): tools/sq_counters.sh <workload> <bytes> <tag>
# (GPU box, repo root).  Output: gpurun_out/<tag>_sq_counters.txt, one line per kernel with the sums over its dispatches.
which=${1:-uniform}; size=${2:-10e9}; tag=${3:-r03_$which}
out=$GRAFT_REPO_ROOT/gpurun_out/sq_$tag; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  printf 'pmc: %s\n' "$set" > $out/in_$n.txt
  rocprofv3 -i $out/in_$n.txt --output-format csv -d $out/raw_$n -- python $GRAFT_REPO_ROOT/tools/perf_side.py $which $size > $out/run_$n.log 2>&1
done
cd $GRAFT_REPO_ROOT
python - "$out" "$tag" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
agg = collections.OrderedDict()
for f in glob.glob(out + "/raw_*/**/*counter_collection.csv", recursive=True):
    seen = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = agg.setdefault(k, collections.OrderedDict())
        a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        seen[(k, r["Counter_Name"])] += 1
    for (k, c), n in seen.items(): agg[k]["_rows_" + c] = n
with open("gpurun_out/%s_sq_counters.txt" % tag, "w") as o:
    for k, a in agg.items():
        o.write(k + " " + " ".join("%s=%d" % (c, v) for c, v in a.items() if not c.startswith("_rows_")) + " dispatch_rows=%d\n" % max(v for c, v in a.items() if c.startswith("_rows_")))
PY
rm -rf $out/raw_*
grep "k_emit_tile_flat" gpurun_out/${tag}_sq_counters.txt
