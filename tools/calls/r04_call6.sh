#!/bin/bash
# round 4, call 6: uniform-tile shortcut with the classifier pre-pass (tile_class_k) instead of the in-kernel vote — parity, then same-box A/B against _ab_old
# (= the commit before the shortcut); general-only run (BSX_NO_UNIFORM_TILES) to see what the shortcut costs where no tile is uniform
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py tests/test_gpu_live.py -x -q -m gpu -k "uniform or mask or blend or step or composite or yuyv or flips or twin or in_place or end_to_end or partial or roi or live or host" 2>&1 | tail -3 | tee gpurun_out/r04_call6_pytest.txt
grep -q "failed\|error" gpurun_out/r04_call6_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
run() { cd $1; env $4 timeout 200 python bench.py --no-cpu-baseline --no-host-io --no-extra-configs --profile-iters 3 --steps 100 --warmup 10 --ramp-seconds 0.5 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}
print('$2', '$3', 'step', d['ms_per_step'], 'fps', d['value'], 'mask_blend', t.get('mask_blend'), d['roofline'].get('tiles'), d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('achieved_dense_10Bpx'))"; }
for i in 1 2; do run $ROOT/_ab_old old ""; run $ROOT new ""; run $ROOT new_general_only "" BSX_NO_UNIFORM_TILES=1; done 2>&1 | tee gpurun_out/r04_call6_ab.txt
for a in "--model mlkit --width 1280 --height 720" "--model full --width 1280 --height 720 --batch 1024" "--model deeplab --batch 1024 --steps 10 --warmup 3" "--model mlkit"; do
  run $ROOT/_ab_old old "$a"; run $ROOT new "$a"
done 2>&1 | tee -a gpurun_out/r04_call6_ab.txt
