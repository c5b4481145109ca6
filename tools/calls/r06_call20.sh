#!/bin/bash
# round 6, call 20: specialisation probe of prep_fused_k and mask_tile_k (lite VGA geometry as constants) vs the release library
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
OUT=gpurun_out/r06s_img_probe_lite.txt; : > $OUT
for rep in 1 2 3 4; do
  for v in "BSX_X=0" "BSX_LIBRARY=$ROOT/backscrub_amd/libbsx_probe_img.so"; do
    env $v timeout 600 python tools/exp_mid_geometry.py --model lite --batch 256 --tag "$(basename ${v#*=})" 2>>gpurun_out/r06s_err.txt | tail -1 >> $OUT
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r06s_img_probe_lite.txt'):
    d=json.loads(l); u=d['launch_us']; print(d['tag'][:22], 'step', d['step_ms'], 'fps', d['fps'], {k:u[k] for k in ('prep','mask_blend')}, d['iou_min'], d['max_abs'])
PY
tail -2 gpurun_out/r06s_err.txt
