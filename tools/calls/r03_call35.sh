#!/bin/bash
# upper bound of hoisting the gate prologue out of seg_k2 / seg_tail (BSX_SEG_GATE_SKIP=1: results invalid, timing only)
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { env $1 timeout 200 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 3 --steps 200 --warmup 20 --ramp-seconds 1 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}; print('$1 $2', round(d['value']), d['ms_per_step'], [t.get(k) for k in ('seg_k2','seg_tail+decode')])"; }
for rep in 1 2; do run X=1 ""; run BSX_SEG_GATE_SKIP=1 ""; done
run X=1 "--model mlkit --width 1280 --height 720 --steps 30 --warmup 5"; run BSX_SEG_GATE_SKIP=1 "--model mlkit --width 1280 --height 720 --steps 30 --warmup 5"
