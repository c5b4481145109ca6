#!/bin/bash
# round 6, call 27: tile launch cut into the rectangle of whole tiles + partial strips (MLKit / DeepLab: square ROI) — parity, then same-box A/B against the previous library
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu -k "mlkit or deeplab or composite or step or mask" 2>&1 | tail -3
run() { BSX_LIBRARY=$ROOT/backscrub_amd/$1 python bench.py $2 --no-extra-configs --no-cpu-baseline --no-side-probes --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', '$3', round(d['value']), d['ms_per_step'], [t for t in d['top_launches'] if t[0]=='mask_blend'], d.get('parity_sample'))"; }
for i in 1 2 3; do
  for L in libbsx_prev.so libbsx.so; do
    run $L "--model mlkit --batch 256 --width 1280 --height 720" mlkit_hd
    run $L "--model deeplab --batch 1024" deeplab
  done
done 2>&1 | tee gpurun_out/r06ac_tile_rect_ab.txt
