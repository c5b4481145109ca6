#!/bin/bash
# round 5, call 8: evidence on the final kernels (tools/calls/r05_call4.sh: kernel stats, HBM counters -> pmc_latest.json, SQ counters, roctx ranges), then the bench as the
# driver runs it and with its defaults
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
bash tools/calls/r05_call4.sh r05z > gpurun_out/r05z_collect.log 2>&1
cp gpurun_out/pmc_latest.json profiles/pmc_latest.json          # (bench.py reads the committed copy; the digest must be the one of THIS tree)
cd $ROOT
S=$(date +%s)
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r05z_bench_driver_detail.json > gpurun_out/r05z_bench_driver.json 2> gpurun_out/r05z_bench_driver.err
echo "driver-style rc=$? wall=$(( $(date +%s) - S )) s bytes=$(wc -c < gpurun_out/r05z_bench_driver.json)" | tee gpurun_out/r05z_bench_driver.txt
S=$(date +%s)
timeout 900 python bench.py --detail gpurun_out/r05z_bench_detail.json > gpurun_out/r05z_bench.json 2> gpurun_out/r05z_bench.err
echo "default rc=$? wall=$(( $(date +%s) - S )) s bytes=$(wc -c < gpurun_out/r05z_bench.json)" | tee -a gpurun_out/r05z_bench_driver.txt
cat gpurun_out/r05z_bench.json
