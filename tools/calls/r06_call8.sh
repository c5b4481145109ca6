#!/bin/bash
# round 6, call 8: plain vs opaque lane index (BSX_RTC_TID=0 / 1, debug build) per model, zero-cell chunked depthwise on in both; then the release build's own choice
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
OUT=$ROOT/gpurun_out/r06g_mid_tid_per_model.txt; : > $OUT
for rep in 1 2 3; do
  for cfg in "--model lite --batch 256" "--model mlkit --batch 256 --width 1280 --height 720 --steps 60" "--model full --batch 1024 --width 1280 --height 720 --steps 40"; do
    for v in "BSX_RTC_TID=0" "BSX_RTC_TID=1"; do
      env BSX_LIBRARY=$ROOT/backscrub_amd/libbsx_dbg.so $v timeout 600 python tools/exp_mid_geometry.py $cfg --tag "$v" 2>>$ROOT/gpurun_out/r06g_err.txt | tail -1 | tee -a $OUT
    done
    timeout 600 python tools/exp_mid_geometry.py $cfg --tag "release" 2>>$ROOT/gpurun_out/r06g_err.txt | tail -1 | tee -a $OUT
  done
done
tail -3 $ROOT/gpurun_out/r06g_err.txt
