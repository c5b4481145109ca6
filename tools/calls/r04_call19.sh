#!/bin/bash
# round-4 evidence (tag r04z): per-config kernel stats + HBM PMC passes, merged counters (pmc_latest.json stamped with the csrc digest), the default bench line over them,
# the whole GPU test suite
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
T0=$SECONDS
timeout 500 bash tools/profile_round.sh r04z > gpurun_out/r04z_profile_round.log 2>&1; echo "profile_round $((SECONDS-T0)) s"
timeout 400 bash tools/profile_config.sh r04z lite '{"batch":256,"width":640,"height":480,"model":"segm_lite_v681.tflite"}' > /dev/null 2>&1
timeout 500 bash tools/profile_config.sh r04z mlkit_hd '{"batch":256,"width":1280,"height":720,"model":"selfiesegmentation_mlkit-256x256-2021_01_19-v1215.f16.tflite"}' --model mlkit --batch 256 --width 1280 --height 720 > /dev/null 2>&1
timeout 700 bash tools/profile_config.sh r04z deeplab '{"batch":1024,"width":640,"height":480,"model":"deeplabv3_257_mv_gpu.tflite"}' --model deeplab --batch 1024 --bg-ring > /dev/null 2>&1
timeout 700 bash tools/profile_config.sh r04z full_hd '{"batch":1024,"width":1280,"height":720,"model":"segm_full_v679.tflite"}' --model full --batch 1024 --width 1280 --height 720 > /dev/null 2>&1
echo "profile_config x4 $((SECONDS-T0)) s"
python tools/merge_pmc.py r04z lite mlkit_hd deeplab full_hd
timeout 900 python bench.py > gpurun_out/r04z_bench.json 2> gpurun_out/r04z_bench.err; echo "bench $((SECONDS-T0)) s"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04z_bench.json'))
print(d['value'], d['ms_per_step'], json.dumps(d.get('roofline'))[:900])
print([(t['name'], t['ms']) for t in d['top_launches']])
print('pipelined', d.get('pipelined', {}).get('value'), 'yuyv', d.get('yuyv_out', {}).get('value'), 'composite_only', d.get('composite_only', {}).get('value'))
for c in d.get('configs', []):
    print(c.get('workload', '')[:50], c.get('value'), c.get('ms_per_step'), (c.get('roofline') or {}).get('traffic'), (c.get('roofline') or {}).get('traffic_stale'), c.get('parity_sample'))
for c in d.get('act_modes', []) + d.get('gemm_modes', []):
    print({k: c.get(k) for k in ('BSX_ACT16', 'BSX_F16_GEMM', 'value', 'ms_per_step', 'error')})
print(d.get('cpu_baseline', {}).get('legs'))
PY
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r04z_pytest.txt 2>&1; tail -4 gpurun_out/r04z_pytest.txt; echo "pytest $((SECONDS-T0)) s"
head -14 gpurun_out/r04z_kernel_stats.md
ls gpurun_out | grep r04z | head -40
