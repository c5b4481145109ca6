#!/bin/bash
# round 5, call 14: middle kernel — out-of-image depthwise taps read a zero cell of the LDS block through an address select (no per-register zeroing):
# parity (every execution path), then same-box A/B of the middle kernel on the three Meet / MLKit configurations
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -4 | tee gpurun_out/r05l_pytest.txt
grep -q "failed\|error" gpurun_out/r05l_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
run() { ( cd $1; timeout 300 python bench.py --no-cpu-baseline --no-host-io --no-extra-configs --no-side-probes --profile-iters 8 --steps 100 --warmup 10 --ramp-seconds 1.0 $3 --detail /tmp/ab_detail.json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('/tmp/ab_detail.json')); t={x['name']:x['ms'] for x in f['top_launches']}
print('$2', '$3', 'step', d['ms_per_step'], 'fps', d['value'], 'frame_program', t.get('frame_program'))" ); }
for cfg in "--model lite" "--model mlkit --width 1280 --height 720" "--model full --width 1280 --height 720 --batch 1024"; do
  for i in 1 2 3; do run $ROOT/_ab_old old "$cfg"; run $ROOT new "$cfg"; done
done 2>&1 | tee gpurun_out/r05l_dw_zero_cell_ab.txt
