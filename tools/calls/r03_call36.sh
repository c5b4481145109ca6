#!/bin/bash
# the tail's gate computed once per frame (seg_gate_k) instead of per tile: whole GPU suite, A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r03ac_pytest.txt 2>&1; tail -3 gpurun_out/r03ac_pytest.txt
run() { env $1 timeout 200 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 3 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}; print('$1 $2', round(d['value']), d['ms_per_step'], [t.get(k) for k in ('seg_k3','seg_gate','seg_tail+decode')], d.get('full_batch_twin_streams',{}).get('all_identical'))"; }
for rep in 1 2; do run X=1 "--steps 200 --warmup 20 --ramp-seconds 1"; run BSX_SEG_NO_GATE_KERNEL=1 "--steps 200 --warmup 20 --ramp-seconds 1"; done
run X=1 "--model mlkit --width 1280 --height 720 --steps 30 --warmup 5 --ramp-seconds 1"; run BSX_SEG_NO_GATE_KERNEL=1 "--model mlkit --width 1280 --height 720 --steps 30 --warmup 5 --ramp-seconds 1"
run X=1 "--model full --batch 1024 --width 1280 --height 720 --steps 20 --warmup 5 --ramp-seconds 1"; run BSX_SEG_NO_GATE_KERNEL=1 "--model full --batch 1024 --width 1280 --height 720 --steps 20 --warmup 5 --ramp-seconds 1"
