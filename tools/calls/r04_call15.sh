#!/bin/bash
# round 4, call 15: placement-policy search of the middle program (segm_full: long-lived skip to the arena, expanded tensors elided, small tensors top-down):
# parity of the new plan, then A/B of the policies on one box (BSX_PLAN_POLICY=0 = the round-3 planner)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -m gpu -q -x -k "full and not deeplab" 2>&1 | tail -4 | tee gpurun_out/r04n_pytest.txt
grep -q "failed\|error" gpurun_out/r04n_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
run() { local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${envs[*]} $*', round(d['value']), d['ms_per_step'], [(t['name'],t['ms']) for t in d['top_launches'] if t['name']=='frame_program'], d.get('full_batch_twin_streams',{}).get('all_identical'))"; }
for rep in 1 2; do
for pol in 0 7; do
run BSX_PLAN_POLICY=$pol -- --model full --batch 1024 --width 1280 --height 720 --steps 20 --warmup 5 --ramp-seconds 1
done; done 2>&1 | tee gpurun_out/r04n_plan_policy_ab.txt
