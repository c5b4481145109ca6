#!/bin/bash
# segment kernels without the quad transpose (weights as the MFMA A operand): parity + timing
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not deeplab and not gaussian" 2>&1 | tail -4
run() { env $1 timeout 200 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 3 --steps 60 --warmup 10 --ramp-seconds 0.5 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}; print('$1 $2', d['ms_per_step'], [t.get(k) for k in ('seg_head','seg_k2','seg_k3','seg_tail+decode')])"; }
run X=0
run X=0
run X=0 "--model mlkit --width 1280 --height 720"
run X=0 "--model full --width 1280 --height 720"
