#!/bin/bash
# round 6, call 16: specialisation probe — the four segment kernels with segm_lite's descriptors as compile-time constants (libbsx_probe_lite.so) vs the release library
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
OUT=gpurun_out/r06o_seg_probe_lite.txt; : > $OUT
for rep in 1 2 3 4; do
  for v in "BSX_X=0" "BSX_LIBRARY=$ROOT/backscrub_amd/libbsx_probe_lite.so"; do
    env $v timeout 600 python tools/exp_mid_geometry.py --model lite --batch 256 --tag "$(basename ${v#*=})" 2>>gpurun_out/r06o_err.txt | tail -1 >> $OUT
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r06o_seg_probe_lite.txt'):
    d=json.loads(l); u=d['launch_us']; print(d['tag'][:22], 'step', d['step_ms'], 'fps', d['fps'], {k:u[k] for k in ('seg_head','seg_k2','frame_program','seg_k3','seg_tail+decode')}, d['iou_min'], d['max_abs'])
PY
tail -2 gpurun_out/r06o_err.txt
