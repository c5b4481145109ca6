#!/bin/bash
# round 4, call 16: seg_head — stem -> 1x1 through registers, x rows of AC pixels (5 workgroups per CU): parity of every Meet / MLKit path, then same-box A/B against _ab_old
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not deeplab" 2>&1 | tail -3 | tee gpurun_out/r04o_pytest.txt
grep -q "failed\|error" gpurun_out/r04o_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
run() { cd $1; timeout 200 python bench.py --no-cpu-baseline --no-host-io --no-extra-configs --profile-iters 3 --steps 100 --warmup 10 --ramp-seconds 0.5 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}
print('$2', '$3', 'step', d['ms_per_step'], 'fps', d['value'], 'seg_head', t.get('seg_head'), 'frame_program', t.get('frame_program'))"; }
for i in 1 2; do run $ROOT/_ab_old old ""; run $ROOT new ""; done 2>&1 | tee gpurun_out/r04o_seg_head_ab.txt
for i in 1 2; do
run $ROOT/_ab_old old "--model mlkit --width 1280 --height 720 --steps 40" 2>&1 | tee -a gpurun_out/r04o_seg_head_ab.txt
run $ROOT new "--model mlkit --width 1280 --height 720 --steps 40" 2>&1 | tee -a gpurun_out/r04o_seg_head_ab.txt
done
run $ROOT/_ab_old old "--model full --width 1280 --height 720 --batch 1024 --steps 20" 2>&1 | tee -a gpurun_out/r04o_seg_head_ab.txt
run $ROOT new "--model full --width 1280 --height 720 --batch 1024 --steps 20" 2>&1 | tee -a gpurun_out/r04o_seg_head_ab.txt
