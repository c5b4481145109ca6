#!/bin/bash
# round 6, call 12: 33x33 fused expand + depthwise layers with 16-channel chunks (70 KB of LDS: two workgroups per CU) instead of 32 (139 KB: one) — debug build, BSX_IR_GEOM
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
OUT=gpurun_out/r06k_ir_geom33.txt; : > $OUT
for rep in 1 2; do
  for v in "BSX_X=0" "BSX_IR_GEOM=33:16,33" "BSX_IR_GEOM=33:24,33"; do
    env BSX_LIBRARY=$ROOT/backscrub_amd/libbsx_dbg.so $v timeout 900 python tools/exp_mid_geometry.py --model deeplab --batch 1024 --steps 12 --tag "$v" 2>>gpurun_out/r06k_err.txt | tail -1 >> $OUT
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r06k_ir_geom33.txt'):
    d=json.loads(l); u=d['launch_us']
    print(d['tag'], 'step', d['step_ms'], 'iou', d['iou_min'], d['max_abs'], {k:u[k] for k in ('conv#50+dw#51','conv#39+dw#40','conv#24+dw#25','conv#13+dw#14') if k in u})
PY
tail -2 gpurun_out/r06k_err.txt
