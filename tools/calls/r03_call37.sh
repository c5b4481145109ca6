#!/bin/bash
# phase-skip upper bounds of the segment kernels (BSX_SEG_SKIP="head,k2,k3,tail" masks; results invalid, timing only), lite/VGA
cd ${GRAFT_REPO_ROOT:-/root/repo}
run() { BSX_SEG_SKIP=$1 timeout 200 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 3 --steps 60 --warmup 10 --ramp-seconds 0.5 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}; print('$1 $2', d['ms_per_step'], [t.get(k) for k in ('seg_head','seg_k2','seg_k3','seg_tail+decode')])"; }
run 0,0,0,0
run 1,0,0,0
run 2,0,0,0
run 4,0,0,0
run 8,0,0,0
run 15,0,0,0
run 0,2,0,0
run 0,0,0,1
run 0,0,0,2
run 0,0,0,3
run 0,0,0,0 "--model mlkit --width 1280 --height 720"
run 15,2,0,3 "--model mlkit --width 1280 --height 720"
