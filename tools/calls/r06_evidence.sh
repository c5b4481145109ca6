#!/bin/bash
# round 6, evidence call (evidence only, on the final kernels): per-workload kernel stats + HBM counters (-> profiles/pmc_latest.json), kernel stats of the default bench
# command, SQ wave-state / LDS counters of the four configurations, and a roctx marker trace of three steps
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-r06z}
cd $ROOT; mkdir -p gpurun_out
timeout 400 bash tools/profile_config.sh $TAG lite '{"batch":256,"width":640,"height":480,"model":"segm_lite_v681.tflite"}' > /dev/null 2>&1
timeout 400 bash tools/profile_config.sh $TAG mlkit_hd '{"batch":256,"width":1280,"height":720,"model":"selfiesegmentation_mlkit-256x256-2021_01_19-v1215.f16.tflite"}' --model mlkit --width 1280 --height 720 > /dev/null 2>&1
timeout 500 bash tools/profile_config.sh $TAG deeplab '{"batch":1024,"width":640,"height":480,"model":"deeplabv3_257_mv_gpu.tflite"}' --model deeplab --batch 1024 --bg-ring > /dev/null 2>&1
timeout 500 bash tools/profile_config.sh $TAG full_hd '{"batch":1024,"width":1280,"height":720,"model":"segm_full_v679.tflite"}' --model full --batch 1024 --width 1280 --height 720 > /dev/null 2>&1
python tools/merge_pmc.py $TAG lite mlkit_hd deeplab full_hd
# the default bench command's own kernel averages (what roofline.avg_ms is checked against)
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_$TAG -o bench -- python $ROOT/bench.py --no-cpu-baseline --no-extra-configs --no-side-probes --profile-iters 4 --ramp-seconds 1.0 > $ROOT/gpurun_out/prof_$TAG.log 2>&1
cd $ROOT
python tools/rocpd_summary.py gpurun_out/prof_$TAG/bench_results.db > gpurun_out/${TAG}_kernel_stats.md; rm -rf gpurun_out/prof_$TAG
grep '^{' gpurun_out/prof_$TAG.log | tail -1 > gpurun_out/${TAG}_kernel_stats_bench_line.json      # (the log's last line is rocprofv3's own)
# SQ counters
timeout 300 bash tools/pmc_sq2.sh ${TAG}_lite > /dev/null 2>&1; cp gpurun_out/pmc_sq2_${TAG}_lite.md gpurun_out/${TAG}_pmc_sq_lite.md
timeout 300 bash tools/pmc_sq2.sh ${TAG}_mlkit --model mlkit --batch 256 --width 1280 --height 720 > /dev/null 2>&1; cp gpurun_out/pmc_sq2_${TAG}_mlkit.md gpurun_out/${TAG}_pmc_sq_mlkit_hd.md
timeout 400 bash tools/pmc_sq2.sh ${TAG}_full --model full --batch 1024 --width 1280 --height 720 > /dev/null 2>&1; cp gpurun_out/pmc_sq2_${TAG}_full.md gpurun_out/${TAG}_pmc_sq_full_hd.md
timeout 400 bash tools/pmc_sq2.sh ${TAG}_deeplab --model deeplab --batch 1024 > /dev/null 2>&1; cp gpurun_out/pmc_sq2_${TAG}_deeplab.md gpurun_out/${TAG}_pmc_sq_deeplab.md
rm -rf gpurun_out/pmc_sq2a_${TAG}_* gpurun_out/pmc_sq2b_${TAG}_*
# roctx ranges (bsx:prep / bsx:network / bsx:mask+blend) next to the kernels they enqueue
cd /tmp
timeout 300 rocprofv3 --marker-trace --kernel-trace --output-format csv -d $ROOT/gpurun_out/roctx_$TAG -o t -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --no-side-probes --no-host-io --profile-iters 1 --ramp-seconds 0 > $ROOT/gpurun_out/roctx_$TAG.log 2>&1
cd $ROOT
python - <<PY > gpurun_out/${TAG}_roctx_ranges.md
import csv, glob, collections
fs = glob.glob("gpurun_out/roctx_$TAG/**/*marker_api_trace.csv", recursive=True)
print("files:", fs)
for f in fs:
    rows = list(csv.DictReader(open(f)))
    print("columns:", list(rows[0].keys()) if rows else [])
    c = collections.Counter(); d = collections.defaultdict(float)
    for r in rows:
        name = r.get("Function") or r.get("Name") or r.get("Message") or str(r)
        c[name] += 1
        try: d[name] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        except Exception: pass
    print("| range | count | total host us | avg host us |\n|---|---:|---:|---:|")
    for k, v in c.most_common(): print("| %s | %d | %.1f | %.2f |" % (k, v, d[k], d[k] / v))
PY
rm -rf gpurun_out/roctx_$TAG
ls -la gpurun_out | grep $TAG
cat gpurun_out/${TAG}_roctx_ranges.md | head -20
