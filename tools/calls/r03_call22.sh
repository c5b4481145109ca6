#!/bin/bash
# op_pw tile order for arena outputs (N-tile fastest): parity on the program paths, A/B against BSX_RTC_NO_NFAST=1, timelines
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "every_execution_path or end_to_end or act16" > gpurun_out/r03q_pytest.txt 2>&1; tail -4 gpurun_out/r03q_pytest.txt
run() { local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${envs[*]} $*', round(d['value']), d['ms_per_step'], [(t['name'],t['ms']) for t in d['top_launches'] if t['name']=='frame_program'])"; }
for rep in 1 2; do
run X=1 -- --model mlkit --width 1280 --height 720 --steps 30 --warmup 5 --ramp-seconds 1
run BSX_RTC_NO_NFAST=1 -- --model mlkit --width 1280 --height 720 --steps 30 --warmup 5 --ramp-seconds 1
done
run X=1 -- --model full --batch 1024 --width 1280 --height 720 --steps 20 --warmup 5 --ramp-seconds 1
run BSX_RTC_NO_NFAST=1 -- --model full --batch 1024 --width 1280 --height 720 --steps 20 --warmup 5 --ramp-seconds 1
run X=1 -- --steps 200 --warmup 20 --ramp-seconds 1
run BSX_RTC_NO_NFAST=1 -- --steps 200 --warmup 20 --ramp-seconds 1
timeout 300 python tools/program_timeline.py mlkit 256 1280 720 2>/dev/null | grep "conv#\|total" | awk '{print $1, $4, $5}' | tr '\n' ';'
