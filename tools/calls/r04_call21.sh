#!/bin/bash
# round 4, call 21: seg_head without its scratch block (the lite tile's workgroup sat exactly on the 32 KB line): parity, same-box A/B against _ab_old
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not deeplab" 2>&1 | tail -3 | tee gpurun_out/r04s_pytest.txt
grep -q "failed\|error" gpurun_out/r04s_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
run() { cd $1; timeout 200 python bench.py --no-cpu-baseline --no-host-io --no-extra-configs --no-side-probes --profile-iters 3 --steps 100 --warmup 10 --ramp-seconds 0.5 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}
print('$2', '$3', 'step', d['ms_per_step'], 'fps', d['value'], 'seg_head', t.get('seg_head'))"; }
for i in 1 2 3; do run $ROOT/_ab_old old ""; run $ROOT new ""; done 2>&1 | tee gpurun_out/r04s_head_scratch_ab.txt
M="--model mlkit --width 1280 --height 720 --steps 40"
run $ROOT/_ab_old old "$M" 2>&1 | tee -a gpurun_out/r04s_head_scratch_ab.txt
run $ROOT new "$M" 2>&1 | tee -a gpurun_out/r04s_head_scratch_ab.txt
