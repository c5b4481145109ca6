#!/bin/bash
# round 6, call 23: geometry A/B of the chained ASPP head on one box (debug library: BSX_CHAIN_FORM)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
L=backscrub_amd/libbsx_dbg.so
timeout 2000 bash tools/ab_deeplab.sh 2 w4p2=$L w8p2=$L,BSX_CHAIN_FORM=82 w8p1=$L,BSX_CHAIN_FORM=81 w4p1=$L,BSX_CHAIN_FORM=41 w16p1=$L,BSX_CHAIN_FORM=161 three=$L,BSX_NO_CHAIN3=1 2>&1 | tee gpurun_out/r06aa_chain_forms.txt
