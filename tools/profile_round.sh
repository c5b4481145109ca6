#!/bin/bash
# One round's committed profile evidence (run ON the GPU box through gpurun):  bash tools/profile_round.sh r01c
#   1. rocprofv3 --kernel-trace --stats of the default bench command            → gpurun_out/prof_<tag>/
#   2. separate PMC passes (FETCH_SIZE, WRITE_SIZE), --kernel-trace only        → gpurun_out/pmc_{fetch,write}_<tag>/
# Summaries are then written by tools/rocpd_summary.py (here, so the box needs no second trip).
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra-configs --no-side-probes --profile-iters 2 --ramp-seconds 0"
# kernel durations: the bench's own command line (clock ramp, 20 + 200 steps), so that the averages are the warmed-up kernels bench.py times live
BS="python $R/bench.py --no-cpu-baseline --no-extra-configs --no-side-probes --profile-iters 2 --ramp-seconds 1.0"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG -o bench -- $BS > $R/gpurun_out/prof_$TAG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch_$TAG -o bench -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $R/gpurun_out/pmc_write_$TAG -o bench -- $B > /dev/null 2>&1
cd $R
python tools/rocpd_summary.py gpurun_out/prof_$TAG/bench_results.db > gpurun_out/${TAG}_kernel_stats.md
python tools/rocpd_summary.py --pmc gpurun_out/pmc_fetch_$TAG/bench_results.db gpurun_out/pmc_write_$TAG/bench_results.db | grep -v "at::\|rocclr" > gpurun_out/${TAG}_pmc_hbm.md
python tools/rocpd_summary.py --pmc-json $TAG gpurun_out/pmc_fetch_$TAG/bench_results.db gpurun_out/pmc_write_$TAG/bench_results.db > gpurun_out/pmc_$TAG.json
ls -la gpurun_out/${TAG}_* gpurun_out/pmc_$TAG.json
