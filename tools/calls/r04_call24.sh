#!/bin/bash
# round 4, call 24: outside_roi_copy_k with 16-byte accesses — parity of every path that has a ROI border, same-box A/B (MLKit/HD, DeepLab), then the round's evidence once more
# over the final kernels (tag r04z)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu -k "roi or end_to_end or flips or yuyv or partial or pipelined or (mlkit and step)" 2>&1 | tail -3 | tee gpurun_out/r04v_pytest.txt
grep -q "failed\|error" gpurun_out/r04v_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
run() { cd $1; timeout 200 python bench.py --no-cpu-baseline --no-host-io --no-extra-configs --no-side-probes --profile-iters 3 --steps 40 --warmup 10 --ramp-seconds 0.5 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}
print('$2', '$3', 'step', d['ms_per_step'], 'fps', d['value'], 'mask_blend', t.get('mask_blend'))"; }
M="--model mlkit --width 1280 --height 720"
for i in 1 2; do run $ROOT/_ab_old old "$M"; run $ROOT new "$M"; done 2>&1 | tee gpurun_out/r04v_outside_roi_ab.txt
cd $ROOT; bash tools/calls/r04_call19.sh
