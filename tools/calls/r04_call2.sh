#!/bin/bash
# round 4, call 2: full GPU suite on the hygiene / aliasing / visibility build, the default bench line (new roofline accounting, host_io, cpu sweep),
# and the N > 1 job end to end with two ranks sharing the box's one GPU (gloo counters: plumbing run)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
echo "cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null) nproc: $(nproc) $(grep -m1 'model name' /proc/cpuinfo)" | tee gpurun_out/r04_call2_host.txt
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r04_call2_pytest.txt
timeout 600 python bench.py > gpurun_out/r04_call2_bench.json 2> gpurun_out/r04_call2_bench.err; tail -c 600 gpurun_out/r04_call2_bench.err; head -c 1500 gpurun_out/r04_call2_bench.json; echo
timeout 600 python bench.py --gpus 2 --steps 40 --warmup 5 --cpu-seconds 6 > gpurun_out/r04_call2_bench_n2.json 2> gpurun_out/r04_call2_bench_n2.err; tail -c 1500 gpurun_out/r04_call2_bench_n2.err; head -c 3000 gpurun_out/r04_call2_bench_n2.json; echo
