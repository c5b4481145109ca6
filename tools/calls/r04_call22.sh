#!/bin/bash
# round 4, call 22: parity of the Meet / MLKit paths with the planner-side up-sampling scales, then the round's evidence again over the final kernels (tag r04z: replaces the earlier r04z files)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not deeplab" 2>&1 | tail -3 | tee gpurun_out/r04t_pytest.txt
grep -q "failed\|error" gpurun_out/r04t_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
bash tools/calls/r04_call19.sh
