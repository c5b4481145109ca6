#!/bin/bash
# round 6, call 10: where does ir_expand_dw_k spend its time?  (debug build: BSX_IR_PHASES = 1 expand only / 2 depthwise only / 3 both)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
OUT=gpurun_out/r06i_ir_phases.txt; : > $OUT
for v in "BSX_IR_PHASES=3" "BSX_IR_PHASES=1" "BSX_IR_PHASES=2" "BSX_IR_PHASES=3"; do
  env BSX_LIBRARY=$ROOT/backscrub_amd/libbsx_dbg.so $v timeout 900 python tools/exp_mid_geometry.py --model deeplab --batch 1024 --steps 10 --tag "$v" 2>>gpurun_out/r06i_err.txt | tail -1 >> $OUT
done
python - <<'PY'
import json
for l in open('gpurun_out/r06i_ir_phases.txt'):
    d=json.loads(l); print(d['tag'], d['step_ms'], {k:v for k,v in d['launch_us'].items() if '+dw' in k or k.startswith('conv#5') or k.startswith('conv#6') or k.startswith('conv#4')})
PY
tail -2 gpurun_out/r06i_err.txt
