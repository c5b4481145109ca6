#!/bin/bash
# SQ instruction-mix counters for every kernel of one bench step (own pass, --kernel-trace only; see the gpurun rules).
# usage (on the GPU box): bash tools/pmc_sq.sh <tag>   → gpurun_out/pmc_sq_<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --kernel-trace \
  -d $R/gpurun_out/pmc_sq_$TAG -o b -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --profile-iters 1 --ramp-seconds 0 > $R/gpurun_out/pmc_sq_$TAG.log 2>&1
python $R/tools/rocpd_summary.py --pmc $R/gpurun_out/pmc_sq_$TAG/b_results.db | grep -v "at::\|elementwise\|fill_k" 
