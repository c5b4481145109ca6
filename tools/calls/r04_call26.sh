#!/bin/bash
# round 4, call 26 (evidence only, no code change): SQ counters (waves, wave cycles, wait / active / VALU / LDS / VMEM instruction counts, MFMA busy) of the FINAL kernels for
# configs[1] and the configs[4] slice, and the per-op timeline of segm_full's middle kernel under the searched placement
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 300 bash tools/pmc_sq2.sh r04x_lite > /dev/null 2>&1; cp gpurun_out/pmc_sq2_r04x_lite.md gpurun_out/r04x_pmc_sq_lite.md
timeout 400 bash tools/pmc_sq2.sh r04x_full --model full --batch 1024 --width 1280 --height 720 > /dev/null 2>&1; cp gpurun_out/pmc_sq2_r04x_full.md gpurun_out/r04x_pmc_sq_full_hd.md
rm -rf gpurun_out/pmc_sq2a_r04x_* gpurun_out/pmc_sq2b_r04x_*
timeout 200 python tools/program_timeline.py full 1024 1280 720 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r04x_program_timeline_full.txt
BSX_PLAN_POLICY=0 timeout 200 python tools/program_timeline.py full 1024 1280 720 2>/dev/null | grep -v amdgpu.ids > gpurun_out/r04x_program_timeline_full_policy0.txt
tail -3 gpurun_out/r04x_program_timeline_full.txt; tail -3 gpurun_out/r04x_program_timeline_full_policy0.txt
wc -l gpurun_out/r04x_*; head -5 gpurun_out/r04x_pmc_sq_lite.md
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
