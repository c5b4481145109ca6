#!/bin/bash
# HBM-traffic counters (+ kernel stats) of ONE bench configuration (run ON the GPU box through gpurun):
#   bash tools/profile_config.sh <tag> <name> '<workload json>' [bench args]
#   e.g. bash tools/profile_config.sh r03 deeplab '{"batch":1024,"width":640,"height":480,"model":"deeplabv3_257_mv_gpu.tflite"}' --model deeplab --batch 1024 --bg-ring
# → gpurun_out/<tag>_<name>_kernel_stats.md, gpurun_out/<tag>_<name>_pmc_hbm.md, gpurun_out/pmc_<tag>_<name>.json
# Separate --pmc passes with --kernel-trace only (see the gpurun rules); FETCH_SIZE and WRITE_SIZE do not fit one pass (TCC slots).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=$1; NAME=$2; DESC=$3; shift 3
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-configs --no-side-probes --profile-iters 1 --ramp-seconds 0 $@"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_${TAG}_$NAME -o bench -- $B > $R/gpurun_out/prof_${TAG}_$NAME.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch_${TAG}_$NAME -o bench -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write_${TAG}_$NAME -o bench -- $B > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/prof_${TAG}_$NAME/bench_results.db > gpurun_out/${TAG}_${NAME}_kernel_stats.md
python tools/rocpd_summary.py --pmc gpurun_out/pmc_fetch_${TAG}_$NAME/bench_results.db gpurun_out/pmc_write_${TAG}_$NAME/bench_results.db | grep -v "at::\|rocclr" > gpurun_out/${TAG}_${NAME}_pmc_hbm.md
python tools/rocpd_summary.py --pmc-json2 $TAG "$DESC" gpurun_out/pmc_fetch_${TAG}_$NAME/bench_results.db gpurun_out/pmc_write_${TAG}_$NAME/bench_results.db > gpurun_out/pmc_${TAG}_$NAME.json
# the raw databases are large: only the summaries travel back
rm -rf gpurun_out/prof_${TAG}_$NAME gpurun_out/pmc_fetch_${TAG}_$NAME gpurun_out/pmc_write_${TAG}_$NAME
ls -la gpurun_out/${TAG}_${NAME}_* gpurun_out/pmc_${TAG}_$NAME.json
