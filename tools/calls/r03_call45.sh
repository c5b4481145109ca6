#!/bin/bash
# packed two-pixel bilateral in prep_fused_k: parity, then same-box A/B against _ab_old (previous commit)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not deeplab and not gaussian and not blur" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deeplab and (prep or stage or execution)" 2>&1 | tail -2
run() { cd $1; timeout 200 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 3 --steps 60 --warmup 10 --ramp-seconds 0.5 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}; print('$3 $2', d['ms_per_step'], [(k, t.get(k)) for k in ('prep','seg_head','seg_k2','mask_blend')])"; }
run $ROOT/_ab_old "" old
run $ROOT "" new
run $ROOT/_ab_old "" old
run $ROOT "" new
run $ROOT/_ab_old "--model mlkit --width 1280 --height 720" old
run $ROOT "--model mlkit --width 1280 --height 720" new
run $ROOT/_ab_old "--model deeplab --batch 1024" old
run $ROOT "--model deeplab --batch 1024" new
