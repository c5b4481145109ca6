#!/bin/bash
# rocprofv3 kernel stats of the DeepLab configuration (configs[3] geometry, 1024 streams): bash tools/profile_deeplab.sh r02b
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --model deeplab --batch 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --profile-iters 1 --ramp-seconds 0"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_dl_$TAG -o bench -- $B > $R/gpurun_out/prof_dl_$TAG.log 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/prof_dl_$TAG/bench_results.db > gpurun_out/${TAG}_deeplab_kernel_stats.md
