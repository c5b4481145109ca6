#!/bin/bash
# round 4, call 12: kernel trace of the synchronous step vs the two-deep pipeline (start / end stamps per dispatch): do the halves overlap, what does each kernel pay, where are the gaps
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
mkdir -p $ROOT/gpurun_out
for cfg in "lite 640 480 256" "mlkit 1280 720 256" "full 1280 720 1024"; do
  tag=$(echo $cfg | cut -d' ' -f1)
  ( cd $ROOT && timeout 300 rocprofv3 --kernel-trace -d $ROOT/gpurun_out/ovl_$tag -o t -- python tools/overlap_trace.py --run $cfg > $ROOT/gpurun_out/ovl_$tag.log 2>&1 )
  DB=$(find $ROOT/gpurun_out/ovl_$tag -name "*.db" | head -1)
  ( cd $ROOT && python tools/overlap_trace.py --summary $DB ) 2>&1 | sed "s|$ROOT/||" | tee $ROOT/gpurun_out/r04k_overlap_$tag.txt
  rm -rf $ROOT/gpurun_out/ovl_$tag
done
