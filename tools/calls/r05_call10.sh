#!/bin/bash
# round 5, call 10: packed blend arithmetic in 55 instead of 67 VALU instructions per four pixels (two v_pk_mad_u16 + (u + (u >> 8)) >> 8; odd bytes by v_perm_b32):
# blend / tile parity (incl. the exhaustive 2^24 byte-triple test), then same-box A/B
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r05h_pytest.txt
grep -q "failed\|error" gpurun_out/r05h_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
run() { ( cd $1; timeout 300 python bench.py --no-cpu-baseline --no-host-io --no-extra-configs --no-side-probes --profile-iters 8 --steps 200 --warmup 20 --ramp-seconds 1.0 $3 --detail /tmp/ab_detail.json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('/tmp/ab_detail.json')); t={x['name']:x['ms'] for x in f['top_launches']}
print('$2', '$3', 'step', d['ms_per_step'], 'fps', d['value'], 'mask_blend', t.get('mask_blend'))" ); }
for cfg in "--model lite" "--model full --width 1280 --height 720 --batch 1024" "--model lite --per-stream-bg"; do
  for i in 1 2 3; do run $ROOT/_ab_old old "$cfg"; run $ROOT new "$cfg"; done
done 2>&1 | tee gpurun_out/r05h_blend_math_ab.txt
