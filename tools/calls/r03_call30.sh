#!/bin/bash
# composite-only step (BSX_STEP_NO_MASK): tests + the bench leg
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_bench_contract.py -m gpu -q -x -k "without_storing or flips or bench" > gpurun_out/r03y_pytest.txt 2>&1; tail -4 gpurun_out/r03y_pytest.txt
for rep in 1 2; do
timeout 300 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 2 --steps 200 --warmup 20 --ramp-seconds 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], d.get('composite_only'), d.get('full_batch_twin_streams',{}).get('all_identical'))"
done
timeout 300 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 2 --model mlkit --width 1280 --height 720 --steps 30 --warmup 5 --ramp-seconds 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'], d.get('composite_only'))"
