#!/bin/bash
# round 6, call 7: middle kernel with the lane index behind an opaque asm per op (no cross-op CSE: MLKit 352 B of spill -> 0, 128 -> 92 registers) + zero-cell taps for the chunked
# depthwise ops — same-box A/B against the previous commit (_ab_old), alternating; then the plain-store switch of the mask tile kernel (debug build)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
OUT=$ROOT/gpurun_out/r06f_mid_tid_ab.txt; : > $OUT
run() { ( cd $1; timeout 600 python tools/exp_mid_geometry.py $3 --tag "$2" 2>>$ROOT/gpurun_out/r06f_err.txt | tail -1 | tee -a $OUT ); }
for rep in 1 2 3; do
  for cfg in "--model lite --batch 256" "--model mlkit --batch 256 --width 1280 --height 720 --steps 60" "--model full --batch 1024 --width 1280 --height 720 --steps 40"; do
    run $ROOT/_ab_old old "$cfg"; run $ROOT new "$cfg"
  done
done
OUT2=$ROOT/gpurun_out/r06f_plain_stores_ab.txt; : > $OUT2
for rep in 1 2 3; do
  for cfg in "--model lite --batch 256" "--model full --batch 1024 --width 1280 --height 720 --steps 40"; do
    for v in "BSX_X=0" "BSX_TILE_PLAIN_STORES=1"; do
      env BSX_LIBRARY=$ROOT/backscrub_amd/libbsx_dbg.so $v timeout 600 python tools/exp_mid_geometry.py $cfg --tag "$v" 2>>$ROOT/gpurun_out/r06f_err.txt | tail -1 | tee -a $OUT2
    done
  done
done
tail -3 $ROOT/gpurun_out/r06f_err.txt
