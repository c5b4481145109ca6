#!/bin/bash
# round 6, call 3: the pruned library (debug switches compiled out, retired paths deleted) — full GPU suite, then the default bench line
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r06c_pytest.txt
timeout 900 python bench.py --no-extra-configs > gpurun_out/r06c_bench.json 2> gpurun_out/r06c_bench.err; tail -c 1500 gpurun_out/r06c_bench.json
