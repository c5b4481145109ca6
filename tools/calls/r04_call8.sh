#!/bin/bash
# round 4, call 8: per-kernel durations (rocprofv3) of the mask + blend launch with and without the uniform-tile shortcut, MLKit/HD and lite/VGA
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for cfg in "lite:" "mlkit_hd:--model mlkit --width 1280 --height 720" "deeplab:--model deeplab --batch 1024"; do
  name=${cfg%%:*}; args=${cfg#*:}
  for mode in on off; do
    envs=""; [ $mode = off ] && envs="BSX_NO_UNIFORM_TILES=1"
    env $envs rocprofv3 --kernel-trace --stats -d $ROOT/gpurun_out/prof_r04g_${name}_$mode -o bench -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-host-io --profile-iters 1 --ramp-seconds 0.3 $args > /dev/null 2>&1
    (cd $ROOT && python tools/rocpd_summary.py gpurun_out/prof_r04g_${name}_$mode/bench_results.db | grep -E "mask_tile_k|tile_class_k|outside_roi|kernel \|" | sed "s/^/$name $mode /")
    rm -rf $ROOT/gpurun_out/prof_r04g_${name}_$mode
  done
done 2>&1 | tee $ROOT/gpurun_out/r04g_mask_tile_kernel_times.txt
