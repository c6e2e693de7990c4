#!/bin/bash
# LDS / wait counters of the Huffman walk (and whatever else runs) in one workload's decode: tools/huf_counters.sh <workload> <bytes> <tag>
# (rocprofv3 --pmc, no tracing domains; GPU box, repo root).  Output: gpurun_out/<tag>_huf_counters.txt
which=${1:-fastq}; size=${2:-4e9}; tag=${3:-r04_$which}
out=$GRAFT_REPO_ROOT/gpurun_out/hc_$tag; mkdir -p $out; cd /tmp; export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_ADDR_CONFLICT" "SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_ACTIVE_INST_VALU"; do
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
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        a = agg.setdefault(k, collections.OrderedDict())
        a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
with open("gpurun_out/%s_huf_counters.txt" % tag, "w") as o:
    for k, a in agg.items():
        if a.get("SQ_WAVE_CYCLES", 0) > 1e8: o.write(k + " " + " ".join("%s=%d" % (c, v) for c, v in a.items()) + "\n")
PY
rm -rf $out/raw_*
grep "k_huf_literals" gpurun_out/${tag}_huf_counters.txt
