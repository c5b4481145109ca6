#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 300 python tools/program_timeline.py lite 256 > gpurun_out/r03f_timeline_lite.txt 2>&1
BSX_RTC_NO_EARLY_FC=1 timeout 300 python tools/program_timeline.py lite 256 > gpurun_out/r03f_timeline_lite_noearly.txt 2>&1
timeout 300 python tools/program_timeline.py full 1024 1280 720 > gpurun_out/r03f_timeline_full.txt 2>&1
timeout 300 python tools/program_timeline.py mlkit 256 1280 720 > gpurun_out/r03f_timeline_mlkit.txt 2>&1
grep total gpurun_out/r03f_*.txt
timeout 900 python -m pytest tests -m gpu -q -x > gpurun_out/r03f_pytest.txt 2>&1
tail -3 gpurun_out/r03f_pytest.txt
timeout 600 python bench.py --no-extra-configs --no-cpu-baseline > gpurun_out/r03f_bench_short.json 2> gpurun_out/r03f_bench_short.err
python -c "
import json; d=json.load(open('gpurun_out/r03f_bench_short.json')); print(d['value'], d['ms_per_step'], [(t['name'],t['ms']) for t in d['top_launches']])"
