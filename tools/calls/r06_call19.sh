#!/bin/bash
# round 6, call 19: mask tile launch cut into whole tile rows (WH instantiation) + the partial last row, for frames whose height is not a multiple of 32 (HD) — parity, then A/B
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu -k "not deeplab" 2>&1 | tail -5 | tee gpurun_out/r06r_pytest.txt
grep -q "failed\|error" gpurun_out/r06r_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
OUT=$ROOT/gpurun_out/r06r_tile_rows_ab.txt; : > $OUT
run() { ( cd $1; timeout 600 python tools/exp_mid_geometry.py $3 --tag "$2" 2>>$ROOT/gpurun_out/r06r_err.txt | tail -1 >> $OUT ); }
for rep in 1 2 3 4; do
  for cfg in "--model full --batch 1024 --width 1280 --height 720 --steps 40" "--model lite --batch 256 --width 1280 --height 720 --steps 60"; do
    run $ROOT/_ab_old old "$cfg"; run $ROOT new "$cfg"
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r06r_tile_rows_ab.txt'):
    d=json.loads(l); print(d['tag'], d['model'], d['batch'], 'mask_blend', d['launch_us']['mask_blend'], 'step', d['step_ms'], 'fps', d['fps'], d['iou_min'], d['max_abs'])
PY
