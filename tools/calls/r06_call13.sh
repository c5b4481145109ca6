#!/bin/bash
# round 6, call 14: mask tile kernel, zero-flags instantiation (F0) + YIN template vs the run-time flags of commit dadc978
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
OUT=$ROOT/gpurun_out/r06m_tile_f0_ab.txt; : > $OUT
run() { ( cd $1; timeout 600 python tools/exp_mid_geometry.py $3 --tag "$2" 2>>$ROOT/gpurun_out/r06l_err.txt | tail -1 >> $OUT ); }
for rep in 1 2 3 4; do
  for cfg in "--model lite --batch 256" "--model full --batch 1024 --width 1280 --height 720 --steps 40"; do
    run $ROOT/_ab_old old "$cfg"; run $ROOT new "$cfg"
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r06m_tile_f0_ab.txt'):
    d=json.loads(l); print(d['tag'], d['model'], d['batch'], 'mask_blend', d['launch_us']['mask_blend'], 'prep', d['launch_us']['prep'], 'step', d['step_ms'], d['iou_min'], d['max_abs'])
PY
