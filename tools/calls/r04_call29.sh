#!/bin/bash
# round 4, call 29: division-free outside-ROI copy — parity of the paths with a ROI border, A/B against the word-indexed kernel on the same build (BSX_NO_COPY16=1)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu -k "roi or flips or yuyv or (end_to_end and mlkit) or (pipelined and mlkit)" 2>&1 | tail -3 | tee gpurun_out/r04y_pytest.txt
grep -q "failed\|error" gpurun_out/r04y_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
run() { env $1 timeout 200 python bench.py --no-cpu-baseline --no-host-io --no-extra-configs --no-side-probes --profile-iters 3 --steps 40 --warmup 10 --ramp-seconds 0.5 --model mlkit --width 1280 --height 720 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}
print('$1', 'step', d['ms_per_step'], 'fps', d['value'], 'mask_blend', t.get('mask_blend'))"; }
for i in 1 2; do run BSX_NO_COPY16=1; run X=1; done 2>&1 | tee gpurun_out/r04y_copy16_ab.txt
