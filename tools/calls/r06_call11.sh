#!/bin/bash
# round 6, call 11: DeepLab pre-split operands — parity (bit-identical to the plan without the hand-over, logits vs oracle at 8 streams), then same-box A/B through the debug switch
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu -k "deeplab" 2>&1 | tail -8 | tee gpurun_out/r06j_pytest.txt
grep -q "failed\|error" gpurun_out/r06j_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
OUT=gpurun_out/r06j_presplit_ab.txt; : > $OUT
for rep in 1 2 3; do
  for v in "BSX_NO_PRESPLIT=1" "BSX_X=0"; do
    env BSX_LIBRARY=$ROOT/backscrub_amd/libbsx_dbg.so $v timeout 900 python tools/exp_mid_geometry.py --model deeplab --batch 1024 --steps 12 --tag "$v" 2>>gpurun_out/r06j_err.txt | tail -1 >> $OUT
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r06j_presplit_ab.txt'):
    d=json.loads(l); u=d['launch_us']
    print(d['tag'], 'step', d['step_ms'], 'fps', d['fps'], 'iou', d['iou_min'], d['max_abs'], {k:u[k] for k in ('conv#50+dw#51','conv#52','conv#39+dw#40','conv#41','conv#24+dw#25','conv#60') if k in u})
PY
tail -2 gpurun_out/r06j_err.txt
