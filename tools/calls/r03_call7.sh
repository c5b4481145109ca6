#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
for rep in 1 2; do
BSX_RTC_FINE=1 timeout 300 python tools/program_timeline.py lite 256 --fine > gpurun_out/r03e_fine_topdown_$rep.txt 2>&1
BSX_PLAN_NO_TOPDOWN=1 BSX_RTC_FINE=1 timeout 300 python tools/program_timeline.py lite 256 --fine > gpurun_out/r03e_fine_notopdown_$rep.txt 2>&1
done
timeout 300 python tools/program_timeline.py lite 256 > gpurun_out/r03e_timeline_lite.txt 2>&1
BSX_PLAN_NO_TOPDOWN=1 timeout 300 python tools/program_timeline.py lite 256 > gpurun_out/r03e_timeline_lite_notopdown.txt 2>&1
grep total gpurun_out/r03e_*.txt
