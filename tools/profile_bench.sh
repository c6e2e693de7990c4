#!/bin/bash
# Produces the files under profiles/ for one round:  tools/profile_bench.sh r02   (run on the GPU box, from the repo root)
#   <tag>_bench_line.json                     the JSON line of a plain `python bench.py`
#   <tag>_bench_rocprofv3_kernel_stats.csv    rocprofv3 --kernel-trace --stats of `python bench.py --steps 3 --no-cpu --softmask-size 0 --realistic-size 0 --fastq1-size 0`
#   <tag>_realistic_rocprofv3_kernel_stats.csv  the same of `tools/perf_side.py realistic` (mostly-flat frame: flat tiles in place, the other blocks decoded beside them)
#   <tag>_pmc_fetch.csv / _pmc_write.csv      per-kernel FETCH_SIZE / WRITE_SIZE sums (separate --pmc passes, no tracing
#                                             domains besides the kernel dispatch records), incl. the calibration kernel
# Everything is written under gpurun_out/<tag>/ ; copy what should be judged into profiles/.
set -u
tag=${1:-r03}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
# the box these files come from (bench.py prints the same stamp as "box" in its line: a `frac` recomputed from the CSVs below and the
# line's own can be matched, or known to come from two boxes)
python -c "import bench; print(bench.box_id())" > $out/${tag}_box.txt
python bench.py > $out/${tag}_bench_line.json 2> $out/bench.err
tail -c 400 $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --steps 3 --no-cpu --softmask-size 0 --realistic-size 0 --fastq1-size 0 > $out/stats.log 2>&1
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/${tag}_bench_rocprofv3_kernel_stats.csv
# the two-pass decode (what archives that are not one flat tree take), every kernel alone on the device: no fused emit, no split
NAF_GPU_FLAT_FUSE=0 NAF_GPU_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats2 -- python bench.py --steps 3 --no-cpu --softmask-size 0 --realistic-size 0 --fastq1-size 0 > $out/stats2.log 2>&1
cp $(ls $out/stats2/*/*kernel_stats.csv | head -1) $out/${tag}_twopass_alone_rocprofv3_kernel_stats.csv
# the serial Huffman kernel on the same data (NAF_GPU_FLAT=0), alone
NAF_GPU_FLAT=0 NAF_GPU_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats3 -- python bench.py --steps 3 --no-cpu --softmask-size 0 --realistic-size 0 --fastq1-size 0 > $out/stats3.log 2>&1
cp $(ls $out/stats3/*/*kernel_stats.csv | head -1) $out/${tag}_serial_huffman_alone_rocprofv3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats4 -- python tools/perf_side.py realistic 4e9 > $out/stats4.log 2>&1
cp $(ls $out/stats4/*/*kernel_stats.csv | head -1) $out/${tag}_realistic_rocprofv3_kernel_stats.csv
bash tools/trace_step.sh uniform 10e9 ${tag}_uniform > /dev/null 2>&1; cp gpurun_out/trace_${tag}_uniform/timeline.txt $out/${tag}_timeline_uniform_10GB.txt; cp gpurun_out/trace_${tag}_uniform/timeline_ennaf.txt $out/${tag}_timeline_ennaf_uniform_10GB.txt
bash tools/trace_step.sh fastq 4e9 ${tag}_fastq > /dev/null 2>&1; cp gpurun_out/trace_${tag}_fastq/timeline.txt $out/${tag}_timeline_fastq_4GB.txt; cp gpurun_out/trace_${tag}_fastq/timeline_ennaf.txt $out/${tag}_timeline_ennaf_fastq_4GB.txt
bash tools/trace_step.sh realistic 4e9 ${tag}_realistic > /dev/null 2>&1; cp gpurun_out/trace_${tag}_realistic/timeline.txt $out/${tag}_timeline_realistic_4GB.txt; cp gpurun_out/trace_${tag}_realistic/timeline_ennaf.txt $out/${tag}_timeline_ennaf_realistic_4GB.txt
python tools/perf_stream.py fastq 4e9 > $out/${tag}_fastq_streams_alone.txt 2>&1
for ctr in FETCH_SIZE WRITE_SIZE; do
  printf 'pmc: %s\n' $ctr > $out/pmc_$ctr.txt
  rocprofv3 -i $out/pmc_$ctr.txt --output-format csv -d $out/pmc_$ctr -- python bench.py --steps 1 --warmup 0 --no-cpu --softmask-size 0 --realistic-size 0 --fastq1-size 0 > $out/pmc_$ctr.log 2>&1
  rocprofv3 -i $out/pmc_$ctr.txt --output-format csv -d $out/cal_$ctr -- tools/bw_calibrate > $out/cal_$ctr.log 2>&1
  NAF_GPU_FLAT=0 NAF_GPU_HUF_PAR=0 rocprofv3 -i $out/pmc_$ctr.txt --output-format csv -d $out/pmcser_$ctr -- python bench.py --steps 1 --warmup 0 --no-cpu --softmask-size 0 --realistic-size 0 --fastq1-size 0 > $out/pmcser_$ctr.log 2>&1
done
python - "$out" "$tag" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
for ctr, name in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    rows = []
    for d, label in ((f"{out}/pmc_{ctr}", "bench"), (f"{out}/cal_{ctr}", "calibration"), (f"{out}/pmcser_{ctr}", "serial_huffman")):
        fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not fs: continue
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] != ctr: continue
            k = r["Kernel_Name"].split("(")[0]
            a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += float(r["Counter_Value"])
        for k, (n, v) in agg.items():
            rows.append((label, k, n, v))
    with open(f"{out}/{tag}_pmc_{name}.csv", "w") as f:
        f.write(f"run,kernel,dispatch_rows,{ctr}_sum\n")
        for r in rows: f.write("%s,%s,%d,%.0f\n" % r)
PY
python - "$out" "$tag" <<'PY'
import csv, json, sys
out, tag = sys.argv[1], sys.argv[2]
line = json.loads(open(f"{out}/{tag}_bench_line.json").read().strip().splitlines()[-1])
text_bytes = int(line["config"]["workload"].split("FASTA ")[-1].split(" B")[0])
# kernel function -> the name bench.py times it under; the PMC passes ran ONE step plus one verification step and one
# instrumented step = 3 unnaf calls, and 2 ennaf calls
names = {"k_huf_literals": "zstd_huf_literals", "k_flat_literals": "zstd_flat_literals", "k_emit_tile": "unnaf_emit", "k_emit_tile_flat": "unnaf_emit_flat",
         "k_emit_rest": "unnaf_emit_rest", "k_build_huf": "zstd_build_huf", "k_spec_find": "zstd_index_find", "k_spec_resolve": "zstd_index_resolve",
         "k_copy_fill": "zstd_copy_fill", "k_flat_streams": "zstd_flat_streams", "k_tile_index": "unnaf_tile_index", "k_stride_probe": "zstd_index_stride", "k_parse_blocks": "zstd_parse_blocks", "k_mask_rle_frame": "unnaf_mask_rle"}
enc_names = {"k_enc_scatter_regular": "ennaf_scatter_regular", "k_enc_scatter": "ennaf_scatter", "k_enc_count_pure": "ennaf_count_pure", "k_enc_count": "ennaf_count", "k_enc_last_fa": "ennaf_last", "k_maskb_count": "ennaf_mask_count", "k_pack_edges_zero": "ennaf_pack_edges",
             "k_zenc_plan": "zenc_plan", "k_zenc_write": "zenc_write", "k_zenc_write_direct": "zenc_write_direct", "k_zenc_flat_scan": "zenc_flat_scan", "k_zenc_tree": "zenc_tree", "k_direct_blocks": "ennaf_direct_blocks",
             "k_mask_run_units": "ennaf_mask_runs", "k_mask_units_write": "ennaf_mask_units"}
calls = 3
enc_calls = 13           # two untimed ennaf calls, ten timed ones and the instrumented one
k = {}
# counter unit = KiB.  FETCH_SIZE tallies a wide coalesced read at 1/2 (guide; k_expand / k_read calibration: x2), the
# Huffman kernel's one-64-byte-sector-per-lane reads at 1/1.742 (k_sector_read calibration); WRITE_SIZE is exact (k_expand / k_write)
fetch_factor = {"k_huf_literals": 1.742}
for ctr, name in (("fetch", "fetch_bytes"), ("write", "write_bytes")):
    for r in csv.DictReader(open(f"{out}/{tag}_pmc_{ctr}.csv")):
        fn = r["kernel"].replace("void ", "").split("<")[0]
        if r["run"] == "serial_huffman" and fn == "k_huf_literals":          # the serial kernel only runs when the flat paths are off: its row comes from that pass
            scale = 1024 * (fetch_factor.get(fn, 2.0) if ctr == "fetch" else 1.0)
            k.setdefault(names[fn], {"fetch_bytes": 0, "write_bytes": 0})[name] += float(r[list(r.keys())[-1]]) * scale / calls
            continue
        if r["run"] != "bench": continue
        if fn in names:
            scale = 1024 * (fetch_factor.get(fn, 2.0) if ctr == "fetch" else 1.0)
            k.setdefault(names[fn], {"fetch_bytes": 0, "write_bytes": 0})[name] += float(r[list(r.keys())[-1]]) * scale / calls
ke = {}
for ctr, name in (("fetch", "fetch_bytes"), ("write", "write_bytes")):
    for r in csv.DictReader(open(f"{out}/{tag}_pmc_{ctr}.csv")):
        if r["run"] != "bench": continue
        fn = r["kernel"].replace("void ", "").split("<")[0]
        if fn in enc_names:
            scale = 1024 * (2.0 if ctr == "fetch" else 1.0)
            ke.setdefault(enc_names[fn], {"fetch_bytes": 0, "write_bytes": 0})[name] += float(r[list(r.keys())[-1]]) * scale / enc_calls
json.dump({"source": f"profiles/{tag}_pmc_fetch.csv + {tag}_pmc_write.csv (rocprofv3 --pmc, separate passes; counter unit KiB; FETCH_SIZE x2, WRITE_SIZE x1; per ennaf call)",
           "text_bytes": text_bytes, "ennaf_calls_in_pass": enc_calls, "kernels": ke}, open(f"{out}/pmc_traffic_ennaf.json", "w"), indent=1)
json.dump({"source": f"profiles/{tag}_pmc_fetch.csv + {tag}_pmc_write.csv (rocprofv3 --pmc, separate passes; counter unit KiB; FETCH_SIZE x2 for coalesced reads, x1.742 for the per-lane 64-byte sector reads of zstd_huf_literals, WRITE_SIZE x1; factors calibrated with tools/bw_calibrate.hip)",
           "text_bytes": text_bytes, "unnaf_calls_in_pass": calls, "kernels": k}, open(f"{out}/pmc_traffic.json", "w"), indent=1)
print(json.dumps(k, indent=1))
PY
ls -la $out/*.csv $out/*.json
