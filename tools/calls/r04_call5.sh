#!/bin/bash
# round 4, call 5: uniform-tile shortcut of the fused mask + blend kernel — parity (both paths, the oracle), then same-box A/B against _ab_old (previous commit)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu -k "uniform or mask or blend or step or composite or yuyv or flips or twin or in_place or end_to_end or partial or roi" 2>&1 | tail -3 | tee gpurun_out/r04_call5_pytest.txt
grep -q "failed\|error" gpurun_out/r04_call5_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
run() { cd $1; env $4 timeout 200 python bench.py --no-cpu-baseline --no-host-io --no-extra-configs --profile-iters 3 --steps 100 --warmup 10 --ramp-seconds 0.5 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}
print('$2', '$3', 'step', d['ms_per_step'], 'fps', d['value'], 'mask_blend', t.get('mask_blend'), d['roofline'].get('tiles'), d['roofline']['kernel'], d['roofline']['frac'], d['roofline'].get('achieved_dense_10Bpx'))"; }
for i in 1 2; do run $ROOT/_ab_old old ""; run $ROOT new ""; done 2>&1 | tee gpurun_out/r04_call5_ab.txt
run $ROOT new_general_only "" BSX_NO_UNIFORM_TILES=1 2>&1 | tee -a gpurun_out/r04_call5_ab.txt
run $ROOT/_ab_old old "--model mlkit --width 1280 --height 720" 2>&1 | tee -a gpurun_out/r04_call5_ab.txt
run $ROOT new "--model mlkit --width 1280 --height 720" 2>&1 | tee -a gpurun_out/r04_call5_ab.txt
run $ROOT/_ab_old old "--model full --width 1280 --height 720 --batch 1024" 2>&1 | tee -a gpurun_out/r04_call5_ab.txt
run $ROOT new "--model full --width 1280 --height 720 --batch 1024" 2>&1 | tee -a gpurun_out/r04_call5_ab.txt
run $ROOT/_ab_old old "--model deeplab --batch 1024 --steps 10 --warmup 3" 2>&1 | tee -a gpurun_out/r04_call5_ab.txt
run $ROOT new "--model deeplab --batch 1024 --steps 10 --warmup 3" 2>&1 | tee -a gpurun_out/r04_call5_ab.txt
