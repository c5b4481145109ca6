#!/bin/bash
# round 6, call 24: chained ASPP head with operand tiles one group ahead and the DMA at the round top — parity + same-box A/B
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "chained_head or batch_of_eight" 2>&1 | tail -3
L=backscrub_amd/libbsx_dbg.so
timeout 2000 bash tools/ab_deeplab.sh 3 top=$L spread=$L,BSX_CHAIN_FORM=420 w16spread=$L,BSX_CHAIN_FORM=1610 2>&1 | tee gpurun_out/r06ab_chain_dma_place.txt
