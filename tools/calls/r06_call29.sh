#!/bin/bash
# round 6, call 29: prep_fused_k with the canvas tile also as floats {R, G, B, 1} in LDS (no byte -> float conversions inside the bilateral filter) — stage tests, then
# same-box A/B against the previous library
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "prep or stage or bilateral or resize or end_to_end or yuyv" 2>&1 | tail -3
run() { BSX_LIBRARY=$ROOT/backscrub_amd/$1 python bench.py $2 --no-extra-configs --no-cpu-baseline --no-side-probes --steps 40 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$3', round(d['value']), d['ms_per_step'], 'prep', d['stage_ms']['prep'])"; }
for i in 1 2 3; do
  for L in libbsx_prev.so libbsx.so; do
    run $L "" lite
    run $L "--model mlkit --batch 256 --width 1280 --height 720" mlkit_hd
  done
done 2>&1 | tee gpurun_out/r06ae_prep_float_tile_ab.txt
