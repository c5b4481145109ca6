#!/bin/bash
# round 5, call 11: mask_tile_k with two tiles per workgroup (BSX_MASK_TILE_PAIRS=1): parity under the switch, then same-build A/B
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
BSX_MASK_TILE_PAIRS=1 timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu -k "end_to_end or roi or flips or yuyv or twin or uniform or full_batch" 2>&1 | tail -3 | tee gpurun_out/r05i_pytest.txt
grep -q "failed\|error" gpurun_out/r05i_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
run() { env $2 timeout 300 python bench.py --no-cpu-baseline --no-host-io --no-extra-configs --no-side-probes --profile-iters 8 --steps 200 --warmup 20 --ramp-seconds 1.0 $3 --detail /tmp/ab_detail.json 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); f=json.load(open('/tmp/ab_detail.json')); t={x['name']:x['ms'] for x in f['top_launches']}
print('$1', '$3', 'step', d['ms_per_step'], 'fps', d['value'], 'mask_blend', t.get('mask_blend'))"; }
for cfg in "--model lite" "--model full --width 1280 --height 720 --batch 1024"; do
  for i in 1 2 3; do run single X=1 "$cfg"; run pairs BSX_MASK_TILE_PAIRS=1 "$cfg"; done
done 2>&1 | tee gpurun_out/r05i_tile_pairs_ab.txt
