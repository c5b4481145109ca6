#!/bin/bash
# round 4, call 1: the prepared DeepLab tail patch (class count / person class as template constants) — parity, then same-box A/B against _ab_old (= round-3 HEAD)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu -k "deeplab" 2>&1 | tail -3 | tee gpurun_out/r04_call1_pytest.txt
grep -q "failed\|error" gpurun_out/r04_call1_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
run() { cd $1; timeout 200 python bench.py --model deeplab --batch 1024 --no-extra-configs --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read())
print('$2', round(d['value']), d['ms_per_step'], [(t['name'], t['ms']) for t in d['top_launches'][:6]], d.get('parity_sample'))"; }
for i in 1 2; do run $ROOT/_ab_old old; run $ROOT new; done 2>&1 | tee gpurun_out/r04_call1_ab.txt
