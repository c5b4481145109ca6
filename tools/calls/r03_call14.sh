#!/bin/bash
# 8-bit network input (the stem normalises on load): parity tests, then A/B against BSX_F32_INPUT=1 on the default job, DeepLab and mlkit/HD
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -m gpu -q -x > gpurun_out/r03j_pytest.txt 2>&1; tail -8 gpurun_out/r03j_pytest.txt
run() { local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${envs[*]} $*', round(d['value']), d['ms_per_step'], [(t['name'],t['ms']) for t in d['top_launches'] if 'prep' in t['name'] or 'head' in t['name'] or 'conv#0' in t['name']])"; }
for rep in 1 2; do
run X=0 -- --steps 100 --warmup 20 --ramp-seconds 1
run BSX_F32_INPUT=1 -- --steps 100 --warmup 20 --ramp-seconds 1
done
for rep in 1 2; do
run X=0 -- --model deeplab --batch 1024 --bg-ring --steps 10 --warmup 3 --ramp-seconds 1
run BSX_F32_INPUT=1 -- --model deeplab --batch 1024 --bg-ring --steps 10 --warmup 3 --ramp-seconds 1
done
run X=0 -- --model mlkit --width 1280 --height 720 --steps 30 --warmup 5 --ramp-seconds 1
run BSX_F32_INPUT=1 -- --model mlkit --width 1280 --height 720 --steps 30 --warmup 5 --ramp-seconds 1
