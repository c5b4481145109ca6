#!/bin/bash
# round 4, call 10 (session 2): health check of the restored tree — full GPU suite + the default bench line at HEAD
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r04i_pytest.txt
timeout 600 python bench.py > gpurun_out/r04i_bench.json 2> gpurun_out/r04i_bench.err; tail -c 600 gpurun_out/r04i_bench.err; head -c 1200 gpurun_out/r04i_bench.json; echo
