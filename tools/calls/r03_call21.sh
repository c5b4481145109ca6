#!/bin/bash
# per-op timelines of the specialised middle program for the two larger graphs (where do MLKit / segm_full spend it?)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 300 python tools/program_timeline.py mlkit 256 1280 720 > gpurun_out/r03p_timeline_mlkit.txt 2>&1
timeout 300 python tools/program_timeline.py full 1024 1280 720 > gpurun_out/r03p_timeline_full.txt 2>&1
timeout 300 python tools/program_timeline.py lite 256 > gpurun_out/r03p_timeline_lite.txt 2>&1
BSX_ACT16=1 timeout 300 python tools/program_timeline.py mlkit 256 1280 720 > gpurun_out/r03p_timeline_mlkit_act16.txt 2>&1
tail -3 gpurun_out/r03p_timeline_*.txt
