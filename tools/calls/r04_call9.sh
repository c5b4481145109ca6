#!/bin/bash
# round 4, call 9: frame-per-workgroup tile classifier — parity of the shortcut, then rocprofv3 per-kernel durations with / without it
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu -k "uniform or mask or blend or step or composite or in_place or end_to_end or partial or roi" 2>&1 | tail -3 | tee gpurun_out/r04_call9_pytest.txt
grep -q "failed\|error" gpurun_out/r04_call9_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
bash tools/calls/r04_call8.sh 2>&1 | sed 's/^/  /'
cp gpurun_out/r04g_mask_tile_kernel_times.txt gpurun_out/r04h_mask_tile_kernel_times.txt
