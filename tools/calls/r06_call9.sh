#!/bin/bash
# round 6, call 9: full GPU suite on the current tree, then the PCIe probe
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r06h_pytest.txt
timeout 600 python tools/host_io_probe.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06h_host_io_probe.txt
