#!/bin/bash
# round 6, call 1: starting state on today's box — GPU tests, the default bench line (chained profile events, 10 B/px tiles, event-overhead removal), FETCH/WRITE calibration
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r06a_pytest.txt
timeout 900 python bench.py > gpurun_out/r06a_bench.json 2> gpurun_out/r06a_bench.err; tail -c 3000 gpurun_out/r06a_bench.json; cp bench_detail.json gpurun_out/r06a_bench_detail.json
bash tools/calibrate_fetch.sh r06a 2>&1 | tail -20
