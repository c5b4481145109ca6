#!/bin/bash
# XCD-aware (frame, tile) order, second batch: mask+blend tile kernels, DeepLab head and tail.  Whole GPU suite, then A/B (BSX_XCD_TILES=0 = plain order everywhere)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r03o_pytest.txt 2>&1; tail -4 gpurun_out/r03o_pytest.txt
run() { local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 2 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${envs[*]} $*', round(d['value']), d['ms_per_step'], [(t['name'],t['ms']) for t in d['top_launches'] if t['name'] in ('mask_blend','prep','conv#0+dw#1+conv#2','resize#69+argmax','seg_tail+decode')])"; }
for rep in 1 2; do
run BSX_XCD_TILES=1 -- --steps 200 --warmup 20 --ramp-seconds 1
run BSX_XCD_TILES=0 -- --steps 200 --warmup 20 --ramp-seconds 1
done
for rep in 1 2; do
run BSX_XCD_TILES=1 -- --model mlkit --width 1280 --height 720 --steps 30 --warmup 5 --ramp-seconds 1
run BSX_XCD_TILES=0 -- --model mlkit --width 1280 --height 720 --steps 30 --warmup 5 --ramp-seconds 1
done
run BSX_XCD_TILES=1 -- --model deeplab --batch 1024 --bg-ring --steps 10 --warmup 3 --ramp-seconds 1
run BSX_XCD_TILES=0 -- --model deeplab --batch 1024 --bg-ring --steps 10 --warmup 3 --ramp-seconds 1
run BSX_XCD_TILES=1 -- --model full --batch 1024 --width 1280 --height 720 --steps 20 --warmup 5 --ramp-seconds 1
run BSX_XCD_TILES=0 -- --model full --batch 1024 --width 1280 --height 720 --steps 20 --warmup 5 --ramp-seconds 1
