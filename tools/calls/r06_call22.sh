#!/bin/bash
# round 6, call 22: the chained ASPP head (pw_chain3_k) — parity, then a same-box A/B against the three GEMM launches (debug library, BSX_NO_CHAIN3)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chained_head or batch_of_eight or every_execution_path or deeplab_argmax" 2>&1 | tail -8 | tee gpurun_out/r06aa_pytest.txt
timeout 1200 bash tools/ab_deeplab.sh 3 chain=backscrub_amd/libbsx_dbg.so three=backscrub_amd/libbsx_dbg.so,BSX_NO_CHAIN3=1 2>&1 | tee gpurun_out/r06aa_chain_ab.txt
