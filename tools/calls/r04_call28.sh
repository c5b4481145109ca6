#!/bin/bash
# round 4, call 28 (evidence only): SQ counters of the final kernels for configs[2] (MLKit/HD) and configs[3] (DeepLab)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 300 bash tools/pmc_sq2.sh r04x_mlkit --model mlkit --batch 256 --width 1280 --height 720 > /dev/null 2>&1; cp gpurun_out/pmc_sq2_r04x_mlkit.md gpurun_out/r04x_pmc_sq_mlkit_hd.md
timeout 400 bash tools/pmc_sq2.sh r04x_deeplab --model deeplab --batch 1024 > /dev/null 2>&1; cp gpurun_out/pmc_sq2_r04x_deeplab.md gpurun_out/r04x_pmc_sq_deeplab.md
rm -rf gpurun_out/pmc_sq2a_r04x_* gpurun_out/pmc_sq2b_r04x_*
wc -l gpurun_out/r04x_pmc_sq_mlkit_hd.md gpurun_out/r04x_pmc_sq_deeplab.md
