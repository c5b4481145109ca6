#!/bin/bash
# middle program: squeeze-excite pool accumulated inside the staged depthwise: parity, A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "every_execution_path or end_to_end or act16 or stages_match" > gpurun_out/r03s_pytest.txt 2>&1; tail -4 gpurun_out/r03s_pytest.txt
run() { local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${envs[*]} $*', round(d['value']), d['ms_per_step'], [(t['name'],t['ms']) for t in d['top_launches'] if t['name']=='frame_program'])"; }
for rep in 1 2; do
run X=1 -- --model mlkit --width 1280 --height 720 --steps 30 --warmup 5 --ramp-seconds 1
run BSX_RTC_NO_DW_POOL=1 -- --model mlkit --width 1280 --height 720 --steps 30 --warmup 5 --ramp-seconds 1
done
run X=1 -- --model full --batch 1024 --width 1280 --height 720 --steps 20 --warmup 5 --ramp-seconds 1
run BSX_RTC_NO_DW_POOL=1 -- --model full --batch 1024 --width 1280 --height 720 --steps 20 --warmup 5 --ramp-seconds 1
