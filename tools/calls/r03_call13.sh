#!/bin/bash
# ACT16 (16-bit activation storage of the segmented networks): GPU tests of the mode, the unchanged default, and A/B timing
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "act16 or every_execution_path or end_to_end" > gpurun_out/r03i_pytest.txt 2>&1; tail -15 gpurun_out/r03i_pytest.txt
run() { local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${envs[*]} $*', round(d['value']), d['ms_per_step'], [(t['name'],t['ms']) for t in d['top_launches']])"; }
for rep in 1 2; do
run BSX_ACT16=0 -- --steps 100 --warmup 20 --ramp-seconds 1
run BSX_ACT16=1 -- --steps 100 --warmup 20 --ramp-seconds 1
done
run BSX_ACT16=0 -- --model mlkit --width 1280 --height 720 --steps 30 --warmup 5 --ramp-seconds 1
run BSX_ACT16=1 -- --model mlkit --width 1280 --height 720 --steps 30 --warmup 5 --ramp-seconds 1
run BSX_ACT16=0 -- --model full --batch 1024 --width 1280 --height 720 --steps 20 --warmup 5 --ramp-seconds 1
run BSX_ACT16=1 -- --model full --batch 1024 --width 1280 --height 720 --steps 20 --warmup 5 --ramp-seconds 1
