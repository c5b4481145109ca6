#!/bin/bash
# exploratory (not the default tree of the round's evidence): DeepLab GEMMs with the weights as the MFMA A operand (no epilogue transposes) — parity, then same-box A/B
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "deeplab" 2>&1 | tail -3
run() { cd $1; timeout 200 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 3 --steps 30 --warmup 5 --ramp-seconds 0.5 --model deeplab --batch 1024 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}; print('$2', d['ms_per_step'], sorted(t.items(), key=lambda kv:-kv[1])[:6])"; }
run $ROOT/_ab_old old
run $ROOT new
run $ROOT/_ab_old old
run $ROOT new
