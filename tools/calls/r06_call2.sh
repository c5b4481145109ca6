#!/bin/bash
# round 6, call 2: middle kernel geometry — 1024 lanes / 160 KB (default) vs 512 / 160 vs 512 / 80 (two workgroups per CU) — same box, alternating, three configurations
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
OUT=gpurun_out/r06b_mid_geometry.txt; : > $OUT
run() { env $1 timeout 600 python tools/exp_mid_geometry.py $2 --tag "$1" 2>>gpurun_out/r06b_err.txt | tail -1 | tee -a $OUT; }
for rep in 1 2; do
  for cfg in "--model lite --batch 256" "--model lite --batch 1024" "--model full --batch 1024 --width 1280 --height 720 --steps 40" "--model mlkit --batch 256 --width 1280 --height 720 --steps 60"; do
    for v in "BSX_X=0" "BSX_MID_LANES=512" "BSX_MID_LANES=512 BSX_MID_LDS_KB=80"; do
      run "$v" "$cfg"
    done
  done
done
tail -5 gpurun_out/r06b_err.txt
