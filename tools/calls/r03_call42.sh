#!/bin/bash
# A/B on ONE box: _ab_old (previous commit, quad-transposed op_pw) vs the working tree (operand-swapped op_pw)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() { cd $1; timeout 200 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 3 --steps 60 --warmup 10 --ramp-seconds 0.5 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}; print('$3 $2', d['ms_per_step'], [(k, t.get(k)) for k in ('frame_program','seg_head','seg_k2','mask_blend')])"; }
run $ROOT/_ab_old "" old
run $ROOT "" new
run $ROOT/_ab_old "" old
run $ROOT "" new
run $ROOT/_ab_old "--model mlkit --width 1280 --height 720" old
run $ROOT "--model mlkit --width 1280 --height 720" new
run $ROOT/_ab_old "--model full --width 1280 --height 720" old
run $ROOT "--model full --width 1280 --height 720" new
