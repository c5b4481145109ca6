#!/bin/bash
# segment tile-size sweep under the XCD-aware order (lite/VGA): "hTR,hTC,k2TR,k2TC,k3TR,k3TC,tTR,tTC"; default 4,14,4,7,16,14,16,14
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
run() { BSX_SEG_TILES=$1 timeout 200 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 3 --steps 100 --warmup 10 --ramp-seconds 0.5 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}; print('$1 $2', round(d['value']), d['ms_per_step'], [t.get(k) for k in ('seg_head','seg_k2','seg_k3','seg_tail+decode')])"; }
run 4,14,4,7,16,14,16,14
run 4,14,4,7,16,14,12,14
run 4,14,4,7,16,14,8,14
run 4,14,4,7,16,14,16,10
run 4,14,4,7,12,14,16,14
run 4,14,4,7,8,14,16,14
run 4,10,4,7,16,14,16,14
run 3,14,4,7,16,14,16,14
run 4,14,3,7,16,14,16,14
run 4,14,4,10,16,14,16,14
run 4,14,2,7,16,14,16,14
run 4,14,4,7,16,14,16,14
