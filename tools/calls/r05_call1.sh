#!/bin/bash
# round 5, call 1: the compact bench line as the driver runs it, the bench contract test, the pipelined-stream fix
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_bench_contract.py tests/test_gpu_batch.py -x -q -m gpu -k "contract or pipelined" 2>&1 | tail -5 | tee gpurun_out/r05a_pytest.txt
S=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r05a_bench_driver.json 2> gpurun_out/r05a_bench_driver.err
echo "rc=$? wall=$(( $(date +%s) - S )) s bytes=$(wc -c < gpurun_out/r05a_bench_driver.json)" | tee gpurun_out/r05a_bench_driver.txt
cp bench_detail.json gpurun_out/r05a_bench_driver_detail.json
tail -c 3000 gpurun_out/r05a_bench_driver.err
cat gpurun_out/r05a_bench_driver.json
