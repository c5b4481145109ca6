#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 300 python tools/twin_diag2.py deeplab 1024 > gpurun_out/r03_twin3.txt 2>&1
timeout 300 python tools/program_timeline.py lite 256 > gpurun_out/r03b_timeline_lite_rtc.txt 2>&1
BSX_NO_RTC=1 timeout 300 python tools/program_timeline.py lite 256 > gpurun_out/r03b_timeline_lite_interp.txt 2>&1
timeout 300 python tools/program_timeline.py full 1024 1280 720 > gpurun_out/r03b_timeline_full_rtc.txt 2>&1
timeout 300 python tools/program_timeline.py mlkit 256 1280 720 > gpurun_out/r03b_timeline_mlkit_rtc.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r03b_pytest.txt 2>&1
tail -5 gpurun_out/r03b_pytest.txt
timeout 600 python bench.py --no-extra-configs --no-cpu-baseline > gpurun_out/r03b_bench_short.json 2> gpurun_out/r03b_bench_short.err
tail -c 1500 gpurun_out/r03b_bench_short.json
