#!/bin/bash
# waves experiment (BSX_WAVE): does running k streams at a time through the whole step (same arena addresses every wave) keep the intermediates in the memory-side cache?
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
run() { # label, env..., --, bench args
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  env "${envs[@]}" timeout 300 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 1 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${envs[*]} $*', round(d['value']), d['ms_per_step'], d.get('parity_sample'))"; }
for w in 0 32 64 128 256; do run BSX_WAVE=$w -- --model deeplab --batch 1024 --bg-ring --steps 10 --warmup 3 --ramp-seconds 1; done
for w in 0 64 128; do run BSX_WAVE=$w -- --steps 100 --warmup 20 --ramp-seconds 1; done
for w in 0 32 64 128; do run BSX_WAVE=$w -- --model mlkit --width 1280 --height 720 --steps 30 --warmup 5 --ramp-seconds 1; done
for w in 0 64 128 256; do run BSX_WAVE=$w -- --model full --batch 1024 --width 1280 --height 720 --steps 20 --warmup 5 --ramp-seconds 1; done
