#!/bin/bash
# round 4, call 13: two-deep pipeline with the composite's occupancy capped (LDS pad) x stream priority, against the synchronous step
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 200 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "pipelined and lite" 2>&1 | tail -3
timeout 600 python tools/pipe_sweep.py lite mlkit full 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04l_pipe_sweep.txt
