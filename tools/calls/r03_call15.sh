#!/bin/bash
# (1) flips folded into the blend: parity test.  (2) lanes experiment: k groups of streams on k HIP streams (overlap the HBM-bound tail with the latency-bound network)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "flips or fused_yuyv" > gpurun_out/r03k_pytest.txt 2>&1; tail -8 gpurun_out/r03k_pytest.txt
run() { local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 1 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${envs[*]} $*', round(d['value']), d['ms_per_step'])"; }
for rep in 1 2; do for l in 1 2 3 4; do run BSX_LANES=$l -- --steps 200 --warmup 20 --ramp-seconds 1; done; done
for l in 1 2 4; do run BSX_LANES=$l -- --model deeplab --batch 1024 --bg-ring --steps 10 --warmup 3 --ramp-seconds 1; done
for l in 1 2 4; do run BSX_LANES=$l -- --model mlkit --width 1280 --height 720 --steps 30 --warmup 5 --ramp-seconds 1; done
for l in 1 2 4; do run BSX_LANES=$l -- --model full --batch 1024 --width 1280 --height 720 --steps 20 --warmup 5 --ramp-seconds 1; done
BSX_LANES=2 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -m gpu -q -x -k "full_batch or every_stream or end_to_end" > gpurun_out/r03k_pytest_lanes.txt 2>&1; tail -4 gpurun_out/r03k_pytest_lanes.txt
