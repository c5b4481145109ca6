#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 300 python tools/program_timeline.py lite 256 > gpurun_out/r03d_timeline_lite.txt 2>&1
BSX_RTC_FINE=1 timeout 300 python tools/program_timeline.py lite 256 --fine > gpurun_out/r03d_timeline_lite_fine.txt 2>&1
timeout 300 python tools/program_timeline.py full 1024 1280 720 > gpurun_out/r03d_timeline_full.txt 2>&1
timeout 300 python tools/program_timeline.py mlkit 256 1280 720 > gpurun_out/r03d_timeline_mlkit.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "execution_path or end_to_end or stages_match or full_batch" > gpurun_out/r03d_pytest.txt 2>&1
tail -3 gpurun_out/r03d_pytest.txt
timeout 600 python bench.py --no-extra-configs --no-cpu-baseline > gpurun_out/r03d_bench_short.json 2> gpurun_out/r03d_bench_short.err
tail -c 600 gpurun_out/r03d_bench_short.json
