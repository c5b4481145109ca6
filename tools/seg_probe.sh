#!/bin/bash
# Specialisation probe of the segment kernels (round 6): dump the descriptors of ONE model's plan (debug build, BSX_SEG_DUMP), compile kernels_seg.hip with them as
# compile-time constants (-DBSX_SEG_PROBE) and link it with the release objects into backscrub_amd/libbsx_probe_<key>.so — valid for that model only, never shipped.
#   bash tools/seg_probe.sh lite models/segm_lite_v681.tflite     (here, on CPU; then BSX_LIBRARY=.../libbsx_probe_lite.so on the GPU box)
set -e
R=$(cd "$(dirname "$0")/.." && pwd); KEY=$1; MODEL=$2
cd $R
INC=$R/backscrub_amd/csrc/build_dbg/seg_probe_$KEY.inc
BSX_LIBRARY=$R/backscrub_amd/libbsx_dbg.so BSX_SEG_DUMP=$INC python -c "from backscrub_amd import api; api.model_describe('$MODEL')" > /dev/null
test -s $INC
C=$R/backscrub_amd/csrc; O=$C/build
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -Wno-unused-result -fvisibility=hidden -fvisibility-inlines-hidden -I $O \
  -DBSX_SEG_PROBE="\"$INC\"" -x hip -c $C/kernels_seg.hip -o $C/build_dbg/kernels_seg_probe_$KEY.o
OBJS=$(ls $O/*.o | grep -v "kernels_seg.hip.o\|bs_maskgen_shim.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -Wl,--version-script=$C/libbsx.map -o $R/backscrub_amd/libbsx_probe_$KEY.so $OBJS $C/build_dbg/kernels_seg_probe_$KEY.o -lz -lpthread -lhiprtc -ldl
ls -la $R/backscrub_amd/libbsx_probe_$KEY.so
