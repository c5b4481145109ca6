#!/bin/bash
# whole-block kernel (ir_block_k) for DeepLab's small-input inverted-residual blocks: parity, then A/B (none / front layers only / all that fit)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -m gpu -q -x -k "deeplab" > gpurun_out/r03l_pytest.txt 2>&1; tail -12 gpurun_out/r03l_pytest.txt
run() { local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${envs[*]} $*', round(d['value']), d['ms_per_step'], [(t['name'],t['ms']) for t in d['top_launches']])"; }
for rep in 1 2; do
run BSX_NO_IR_BLOCK=1 -- --model deeplab --batch 1024 --bg-ring --steps 10 --warmup 3 --ramp-seconds 1 --dump-launches gpurun_out/r03l_launches_noblock.txt
run BSX_IR_BLOCK_MINW=60 -- --model deeplab --batch 1024 --bg-ring --steps 10 --warmup 3 --ramp-seconds 1
run X=1 -- --model deeplab --batch 1024 --bg-ring --steps 10 --warmup 3 --ramp-seconds 1 --dump-launches gpurun_out/r03l_launches_block.txt
done
