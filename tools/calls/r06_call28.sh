#!/bin/bash
# round 6, call 28: chained ASPP head with ALL of a wave's input requested up front — parity + same-box A/B against the previous library
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chained_head or batch_of_eight" 2>&1 | tail -3
timeout 2000 bash tools/ab_deeplab.sh 3 prev=backscrub_amd/libbsx_prev.so new=backscrub_amd/libbsx.so 2>&1 | grep -v "inside the chained" | tee gpurun_out/r06ad_chain_input_upfront.txt
