#!/bin/bash
# round-3 evidence: per-config kernel stats + HBM PMC passes, merged counters, the default bench line over them, the whole GPU test suite
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
T0=$SECONDS
timeout 500 bash tools/profile_round.sh r03z > gpurun_out/r03z_profile_round.log 2>&1; echo "profile_round $((SECONDS-T0)) s"
timeout 400 bash tools/profile_config.sh r03z lite '{"batch":256,"width":640,"height":480,"model":"segm_lite_v681.tflite"}' > /dev/null 2>&1
timeout 500 bash tools/profile_config.sh r03z mlkit_hd '{"batch":256,"width":1280,"height":720,"model":"selfiesegmentation_mlkit-256x256-2021_01_19-v1215.f16.tflite"}' --model mlkit --batch 256 --width 1280 --height 720 > /dev/null 2>&1
timeout 700 bash tools/profile_config.sh r03z deeplab '{"batch":1024,"width":640,"height":480,"model":"deeplabv3_257_mv_gpu.tflite"}' --model deeplab --batch 1024 --bg-ring > /dev/null 2>&1
timeout 700 bash tools/profile_config.sh r03z full_hd '{"batch":1024,"width":1280,"height":720,"model":"segm_full_v679.tflite"}' --model full --batch 1024 --width 1280 --height 720 > /dev/null 2>&1
echo "profile_config x4 $((SECONDS-T0)) s"
python tools/merge_pmc.py r03z lite mlkit_hd deeplab full_hd
timeout 900 python bench.py > gpurun_out/r03z_bench.json 2> gpurun_out/r03z_bench.err; echo "bench $((SECONDS-T0)) s"
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03z_bench.json'))
print(d['value'], d['ms_per_step'], d.get('roofline'))
print([(t['name'], t['ms']) for t in d['top_launches']])
for c in d.get('configs', []):
    print(c.get('workload', '')[:50], c.get('value'), c.get('ms_per_step'), (c.get('roofline') or {}).get('traffic'), c.get('parity_sample'))
for c in d.get('act_modes', []) + d.get('gemm_modes', []):
    print({k: c.get(k) for k in ('BSX_ACT16', 'BSX_F16_GEMM', 'workload', 'value', 'ms_per_step', 'parity_sample', 'error')})
print(d.get('cpu_baseline', {}).get('legs'))
PY
ls gpurun_out | head -40
