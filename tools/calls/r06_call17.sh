#!/bin/bash
# round 6, call 17: graph-specialised segment kernels (hipRTC) — full GPU suite (the product path now runs them), then same-box A/B against the ahead-of-time kernels (debug build, BSX_NO_SEG_RTC)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r06p_pytest.txt
grep -q "failed\|error" gpurun_out/r06p_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
OUT=gpurun_out/r06p_seg_rtc_ab.txt; : > $OUT
for rep in 1 2 3; do
  for cfg in "--model lite --batch 256" "--model mlkit --batch 256 --width 1280 --height 720 --steps 60" "--model full --batch 1024 --width 1280 --height 720 --steps 40"; do
    for v in "BSX_NO_SEG_RTC=1" "BSX_X=0"; do
      env BSX_LIBRARY=$ROOT/backscrub_amd/libbsx_dbg.so $v timeout 600 python tools/exp_mid_geometry.py $cfg --tag "$v" 2>>gpurun_out/r06p_err.txt | tail -1 >> $OUT
    done
  done
done
python - <<'PY'
import json
for l in open('gpurun_out/r06p_seg_rtc_ab.txt'):
    d=json.loads(l); u=d['launch_us']; print(d['tag'][:16], d['model'], 'step', d['step_ms'], 'fps', d['fps'], {k:u[k] for k in ('seg_head','seg_k2','seg_k3','seg_tail+decode')}, d['iou_min'], d['max_abs'])
PY
tail -2 gpurun_out/r06p_err.txt
