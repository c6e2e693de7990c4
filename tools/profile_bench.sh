#!/bin/bash
# Produces the files under profiles/ for one round:  tools/profile_bench.sh r06   (run on the GPU box, from the repo root)
#   <tag>_bench_line.json                     the JSON line of a plain `python bench.py` (100 GB headline, every side leg)
#   <tag>_bench_rocprofv3_kernel_stats.csv    rocprofv3 --kernel-trace --stats of the headline alone: `python bench.py --steps 3 --no-cpu` + SIDE0 (below)
#   <tag>_cfg10_rocprofv3_kernel_stats.csv    the same at BASELINE configs[1]'s 10 GB (`--size 10e9`)
#   pmc_traffic_100gb.json / pmc_traffic.json / pmc_traffic_ennaf(_100gb).json   PMC traffic per kernel and call at 100 GB and at 10 GB
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
SIDE0="--no-cpu --softmask-size 0 --realistic-size 0 --fastq1-size 0 --levels-size 0 --cfg10-size 0"     # the headline leg alone
python bench.py > $out/${tag}_bench_line.json 2> $out/bench.err
tail -c 400 $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py --steps 3 $SIDE0 > $out/stats.log 2>&1
cp $(ls $out/stats/*/*kernel_stats.csv | head -1) $out/${tag}_bench_rocprofv3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats10 -- python bench.py --steps 3 --size 10e9 $SIDE0 > $out/stats10.log 2>&1
cp $(ls $out/stats10/*/*kernel_stats.csv | head -1) $out/${tag}_cfg10_rocprofv3_kernel_stats.csv
# the two-pass decode (what archives that are not one flat tree take), every kernel alone on the device: no fused emit, no split
NAF_GPU_FLAT_FUSE=0 NAF_GPU_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats2 -- python bench.py --steps 3 --size 10e9 $SIDE0 > $out/stats2.log 2>&1
cp $(ls $out/stats2/*/*kernel_stats.csv | head -1) $out/${tag}_twopass_alone_rocprofv3_kernel_stats.csv
# the serial Huffman kernel on the same data (NAF_GPU_FLAT=0), alone
NAF_GPU_FLAT=0 NAF_GPU_SPLIT=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats3 -- python bench.py --steps 3 --size 10e9 $SIDE0 > $out/stats3.log 2>&1
cp $(ls $out/stats3/*/*kernel_stats.csv | head -1) $out/${tag}_serial_huffman_alone_rocprofv3_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats4 -- python tools/perf_side.py realistic 4e9 > $out/stats4.log 2>&1
cp $(ls $out/stats4/*/*kernel_stats.csv | head -1) $out/${tag}_realistic_rocprofv3_kernel_stats.csv
# the FASTQ leg (one GPU's share of configs[4]): kernel stats, and below its PMC traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats5 -- python tools/perf_side.py fastq 12.5e9 > $out/stats5.log 2>&1
cp $(ls $out/stats5/*/*kernel_stats.csv | head -1) $out/${tag}_fastq_rocprofv3_kernel_stats.csv
bash tools/trace_step.sh uniform 10e9 ${tag}_uniform > /dev/null 2>&1; cp gpurun_out/trace_${tag}_uniform/timeline.txt $out/${tag}_timeline_uniform_10GB.txt; cp gpurun_out/trace_${tag}_uniform/timeline_ennaf.txt $out/${tag}_timeline_ennaf_uniform_10GB.txt
bash tools/trace_step.sh fastq 12.5e9 ${tag}_fastq > /dev/null 2>&1; cp gpurun_out/trace_${tag}_fastq/timeline.txt $out/${tag}_timeline_fastq_12GB.txt; cp gpurun_out/trace_${tag}_fastq/timeline_ennaf.txt $out/${tag}_timeline_ennaf_fastq_12GB.txt
bash tools/trace_step.sh realistic 4e9 ${tag}_realistic > /dev/null 2>&1; cp gpurun_out/trace_${tag}_realistic/timeline.txt $out/${tag}_timeline_realistic_4GB.txt; cp gpurun_out/trace_${tag}_realistic/timeline_ennaf.txt $out/${tag}_timeline_ennaf_realistic_4GB.txt
python tools/perf_stream.py fastq 12.5e9 2>&1 | grep -v amdgpu.ids > $out/${tag}_fastq_streams_alone.txt
python tools/perf_refstream.py 2e9 fastq 2>&1 | grep -v amdgpu.ids > $out/${tag}_reference_fastq_streams_alone.txt
python tools/perf_exec.py 1e9 2>&1 | grep -v amdgpu.ids > $out/${tag}_exec_modes.txt
for sz in 10e9 100e9; do
for ctr in FETCH_SIZE WRITE_SIZE; do
  printf 'pmc: %s\n' $ctr > $out/pmc_$ctr.txt
  rocprofv3 -i $out/pmc_$ctr.txt --output-format csv -d $out/pmc_${ctr}_$sz -- python bench.py --steps 1 --warmup 0 --size $sz $SIDE0 > $out/pmc_${ctr}_$sz.log 2>&1
  if [ $sz = 10e9 ]; then
    rocprofv3 -i $out/pmc_$ctr.txt --output-format csv -d $out/cal_$ctr -- tools/bw_calibrate > $out/cal_$ctr.log 2>&1
    NAF_GPU_FLAT=0 NAF_GPU_HUF_PAR=0 rocprofv3 -i $out/pmc_$ctr.txt --output-format csv -d $out/pmcser_$ctr -- python bench.py --steps 1 --warmup 0 --size 10e9 $SIDE0 > $out/pmcser_$ctr.log 2>&1
  fi
done
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 -i $out/pmc_$ctr.txt --output-format csv -d $out/pmcfq_$ctr -- python tools/perf_side.py fastq 12.5e9 > $out/pmcfq_$ctr.log 2>&1
done
python - "$out" "$tag" <<'PY'
# the FASTQ leg's traffic: per-kernel sums of the two passes, the calls counted by the kernels that run once per call
import csv, glob, json, sys, collections
out, tag = sys.argv[1], sys.argv[2]
enc_prefix = ("k_enc", "k_fq_", "k_encq", "k_zenc", "k_lz_parse", "k_lz_seqenc", "k_lz_choose", "k_ldm", "k_maskb", "k_mask_", "k_len_unit", "k_pack_edges", "k_need_list", "k_put_bytes", "k_sniff", "k_collect4", "k_add_u64")
res = {"ennaf": collections.OrderedDict(), "unnaf": collections.OrderedDict(), "shared": collections.OrderedDict()}
calls = {"ennaf": 0, "unnaf": 0}
for ctr, fac in (("FETCH_SIZE", 2.0), ("WRITE_SIZE", 1.0)):
    fs = glob.glob(f"{out}/pmcfq_{ctr}/**/*counter_collection.csv", recursive=True)
    if not fs: continue
    n_sniff = n_parse = 0
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] != ctr: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if k == "k_sniff": n_sniff += 1
        if k == "k_parse_container": n_parse += 1
        if not k.startswith(("k_", "__amd")): continue                     # (the generator's and the comparison's torch kernels)
        side = "ennaf" if k.startswith(enc_prefix) else ("shared" if k.startswith(("k_scan_", "k_small_to_host", "__amd")) else "unnaf")
        f = 1.742 if (ctr == "FETCH_SIZE" and k == "k_huf_literals") else fac
        res[side][k] = res[side].get(k, 0.0) + float(r["Counter_Value"]) * 1024 * f
    calls = {"ennaf": n_sniff, "unnaf": n_parse}
if calls["ennaf"] and calls["unnaf"]:
    doc = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `python tools/perf_side.py fastq 12.5e9`; counter unit KiB; FETCH_SIZE x2 (x1.742 for k_huf_literals), WRITE_SIZE x1; per call = sum / calls (k_sniff / k_parse_container dispatches); kernels both directions use (scans, read-backs) are listed apart and in neither total",
           "calls_in_pass": calls}
    for side in ("ennaf", "unnaf"):
        per = {k: int(v / calls[side]) for k, v in sorted(res[side].items(), key=lambda x: -x[1])}
        doc[side] = {"call_traffic_bytes": int(sum(per.values())), "kernels": dict(list(per.items())[:16])}
    doc["shared_bytes_in_pass"] = int(sum(res["shared"].values()))
    json.dump(doc, open(f"{out}/pmc_traffic_fastq.json", "w"), indent=1)
    print("fastq traffic per call:", doc["ennaf"]["call_traffic_bytes"], doc["unnaf"]["call_traffic_bytes"])
PY
python - "$out" "$tag" <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
for ctr, name in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
    rows = []
    for d, label in ((f"{out}/pmc_{ctr}_10e9", "bench"), (f"{out}/pmc_{ctr}_100e9", "bench100"), (f"{out}/cal_{ctr}", "calibration"), (f"{out}/pmcser_{ctr}", "serial_huffman")):
        fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
        if not fs: continue
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(fs[0])):
            if r["Counter_Name"] != ctr: continue
            k = r["Kernel_Name"].split("(")[0].replace(",", ";")          # (template arguments: no commas inside a CSV field)
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
# kernel function -> the name bench.py times it under; a PMC pass ran ONE step plus one verification step and one instrumented step
# = 3 unnaf calls, and 13 ennaf calls (two untimed, ten timed, the instrumented one)
names = {"k_huf_literals": "zstd_huf_literals", "k_flat_literals": "zstd_flat_literals", "k_emit_tile": "unnaf_emit", "k_emit_tile_wave": "unnaf_emit", "k_emit_tile_flat": "unnaf_emit_flat", "k_emit_tile_flat_wave": "unnaf_emit_flat",
         "k_emit_rest": "unnaf_emit_rest", "k_build_huf": "zstd_build_huf", "k_spec_find": "zstd_index_find", "k_spec_resolve": "zstd_index_resolve",
         "k_copy_fill": "zstd_copy_fill", "k_flat_streams": "zstd_flat_streams", "k_uni_streams": "zstd_flat_uniform", "k_uni_head": "zstd_flat_uniform", "k_tile_index": "unnaf_tile_index", "k_stride_probe": "zstd_index_stride", "k_parse_blocks": "zstd_parse_blocks", "k_mask_rle_frame": "unnaf_mask_rle"}
enc_names = {"k_enc_fused": "ennaf_split_once", "k_zenc_write_direct_loc": "zenc_write_direct", "k_direct_verdict": "ennaf_direct_blocks", "k_sparse_list": "ennaf_sparse_list", "k_pure_check": "ennaf_pure_check",
             "k_zenc_direct_plans": "zenc_direct_plans", "k_irregular_list": "ennaf_irregular_list", "k_enc_scatter_regular_list": "ennaf_scatter_regular", "k_need_list": "ennaf_need_list", "k_scan_tile_reduce": "scan", "k_scan_tile_apply": "scan", "k_scan_small": "scan",
             "k_enc_scatter_regular": "ennaf_scatter_regular", "k_enc_scatter": "ennaf_scatter", "k_enc_count_pure": "ennaf_count_pure", "k_enc_count": "ennaf_count", "k_enc_last_fa": "ennaf_last", "k_maskb_count": "ennaf_mask_count", "k_pack_edges_zero": "ennaf_pack_edges",
             "k_zenc_plan": "zenc_plan", "k_zenc_write": "zenc_write", "k_zenc_write_direct": "zenc_write_direct", "k_zenc_flat_scan": "zenc_flat_scan", "k_zenc_tree": "zenc_tree", "k_direct_blocks": "ennaf_direct_blocks",
             "k_mask_run_units": "ennaf_mask_runs", "k_mask_units_write": "ennaf_mask_units"}
calls, enc_calls = 3, 13
# counter unit = KiB.  FETCH_SIZE tallies a wide coalesced read at 1/2 (guide; k_expand / k_read calibration: x2), the
# Huffman kernel's one-64-byte-sector-per-lane reads at 1/1.742 (k_sector_read calibration); WRITE_SIZE is exact (k_expand / k_write)
fetch_factor = {"k_huf_literals": 1.742}
def tables(run):
    k, ke = {}, {}
    for ctr, name in (("fetch", "fetch_bytes"), ("write", "write_bytes")):
        for r in csv.DictReader(open(f"{out}/{tag}_pmc_{ctr}.csv")):
            fn = r["kernel"].replace("void ", "").split("<")[0]
            v = float(r[list(r.keys())[-1]])
            if run == "bench" and r["run"] == "serial_huffman" and fn == "k_huf_literals":    # the serial kernel only runs when the flat paths are off: its row comes from that pass
                k.setdefault(names[fn], {"fetch_bytes": 0, "write_bytes": 0})[name] += v * 1024 * (fetch_factor[fn] if ctr == "fetch" else 1.0) / calls
                continue
            if r["run"] != run: continue
            if fn in names: k.setdefault(names[fn], {"fetch_bytes": 0, "write_bytes": 0})[name] += v * 1024 * (fetch_factor.get(fn, 2.0) if ctr == "fetch" else 1.0) / calls
            if fn in enc_names: ke.setdefault(enc_names[fn], {"fetch_bytes": 0, "write_bytes": 0})[name] += v * 1024 * (2.0 if ctr == "fetch" else 1.0) / enc_calls
    return k, ke
src = f"profiles/{tag}_pmc_fetch.csv + {tag}_pmc_write.csv (rocprofv3 --pmc, separate passes of `python bench.py --steps 1 --warmup 0 --size S` with the side legs off; counter unit KiB; FETCH_SIZE x2 for coalesced reads, x1.742 for the per-lane 64-byte sector reads of zstd_huf_literals, WRITE_SIZE x1; factors calibrated with tools/bw_calibrate.hip)"
for run, size_name, suffix in (("bench", "cfg10", ""), ("bench100", "headline", "_100gb")):
    k, ke = tables(run)
    if not k and not ke: continue
    tb = int(line["roofline"]["text_bytes"]) if run == "bench100" else int((line.get("cfg10") or {}).get("text_bytes") or 0)
    if not tb: continue
    json.dump({"source": src + " -- rows `%s`; per ennaf call" % run, "text_bytes": tb, "ennaf_calls_in_pass": enc_calls, "kernels": ke}, open(f"{out}/pmc_traffic_ennaf{suffix}.json", "w"), indent=1)
    json.dump({"source": src + " -- rows `%s`; per unnaf call" % run, "text_bytes": tb, "unnaf_calls_in_pass": calls, "kernels": k}, open(f"{out}/pmc_traffic{suffix}.json", "w"), indent=1)
    print(run, tb, json.dumps({n: int(v["fetch_bytes"] + v["write_bytes"]) for n, v in k.items()}))
PY
ls -la $out/*.csv $out/*.json
