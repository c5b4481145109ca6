#!/bin/bash
# Pipe-activity counters for the DeepLab kernels (two --pmc passes, --kernel-trace only):  bash tools/pmc_phase.sh <tag> [ENV=VAL ...]
R=${GRAFT_REPO_ROOT:-/root/repo}; TAG=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
B="env $@ python $R/bench.py --model deeplab --batch 1024 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --profile-iters 1 --ramp-seconds 0"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_MISC --kernel-trace \
  -d $R/gpurun_out/pmc_pa_$TAG -o b -- $B > $R/gpurun_out/pmc_p_$TAG.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA --kernel-trace \
  -d $R/gpurun_out/pmc_pb_$TAG -o b -- $B >> $R/gpurun_out/pmc_p_$TAG.log 2>&1
python $R/tools/rocpd_summary.py --pmc $R/gpurun_out/pmc_pa_$TAG/b_results.db $R/gpurun_out/pmc_pb_$TAG/b_results.db | grep -v "at::\|elementwise\|fill_k\|rocclr" > $R/gpurun_out/pmc_phase_$TAG.md
