#!/bin/bash
# round 4, call 20: ragged channel chunks + level-3 88-channel pair fused (segm_full): parity of segm_full / MLKit, same-box A/B against _ab_old (= the r04z evidence commit)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch.py -x -q -m gpu -k "(full or mlkit) and not deeplab" 2>&1 | tail -3 | tee gpurun_out/r04r_pytest.txt
grep -q "failed\|error" gpurun_out/r04r_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
run() { cd $1; timeout 200 python bench.py --no-cpu-baseline --no-host-io --no-extra-configs --no-side-probes --profile-iters 3 --steps 40 --warmup 10 --ramp-seconds 0.5 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}
print('$2', '$3', 'step', d['ms_per_step'], 'fps', d['value'], 'frame_program', t.get('frame_program'))"; }
F="--model full --width 1280 --height 720 --batch 1024 --steps 20"
for i in 1 2; do run $ROOT/_ab_old old "$F"; run $ROOT new "$F"; done 2>&1 | tee gpurun_out/r04r_ragged_chunks_ab.txt
M="--model mlkit --width 1280 --height 720"
run $ROOT/_ab_old old "$M" 2>&1 | tee -a gpurun_out/r04r_ragged_chunks_ab.txt
run $ROOT new "$M" 2>&1 | tee -a gpurun_out/r04r_ragged_chunks_ab.txt
