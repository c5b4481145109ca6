#!/bin/bash
# HBM traffic of the middle program in the 16-bit activation storage mode (lite/VGA and mlkit/HD): FETCH_SIZE / WRITE_SIZE passes
cd /tmp && export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p $R/gpurun_out
for cfg in "lite:" "mlkit_hd:--model mlkit --width 1280 --height 720"; do
  name=${cfg%%:*}; args=${cfg#*:}
  B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-configs --profile-iters 1 --ramp-seconds 0 $args"
  BSX_ACT16=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_f16_$name -o bench -- $B > /dev/null 2>&1
  BSX_ACT16=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_w16_$name -o bench -- $B > /dev/null 2>&1
  python $R/tools/rocpd_summary.py --pmc $R/gpurun_out/pmc_f16_$name/bench_results.db $R/gpurun_out/pmc_w16_$name/bench_results.db | grep -v "at::\|rocclr" > $R/gpurun_out/r03w_act16_${name}_pmc_hbm.md
  rm -rf $R/gpurun_out/pmc_f16_$name $R/gpurun_out/pmc_w16_$name
  grep "bsx_mid\|seg_" $R/gpurun_out/r03w_act16_${name}_pmc_hbm.md
done
