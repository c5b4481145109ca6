#!/bin/bash
# round 4, call 17: k3 / tail — the low-resolution window reserved at what the geometry needs (k3 4 workgroups per CU instead of 3, tail 5 instead of 4 with a 5-waves-per-SIMD
# register budget): parity, same-box A/B against _ab_old (= the seg_head commit); then the head's tile width re-swept now that its LDS cliff moved
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not deeplab" 2>&1 | tail -3 | tee gpurun_out/r04p_pytest.txt
grep -q "failed\|error" gpurun_out/r04p_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
run() { cd $1; env $4 timeout 200 python bench.py --no-cpu-baseline --no-host-io --no-extra-configs --profile-iters 3 --steps 100 --warmup 10 --ramp-seconds 0.5 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}
print('$2', '$3', '$4', 'step', d['ms_per_step'], 'fps', d['value'], 'head', t.get('seg_head'), 'k3', t.get('seg_k3'), 'tail', t.get('seg_tail+decode'))"; }
for i in 1 2; do run $ROOT/_ab_old old ""; run $ROOT new ""; done 2>&1 | tee gpurun_out/r04p_k3_tail_ab.txt
for i in 1 2; do
run $ROOT/_ab_old old "--model mlkit --width 1280 --height 720 --steps 40" 2>&1 | tee -a gpurun_out/r04p_k3_tail_ab.txt
run $ROOT new "--model mlkit --width 1280 --height 720 --steps 40" 2>&1 | tee -a gpurun_out/r04p_k3_tail_ab.txt
done
run $ROOT/_ab_old old "--model full --width 1280 --height 720 --batch 1024 --steps 20" 2>&1 | tee -a gpurun_out/r04p_k3_tail_ab.txt
run $ROOT new "--model full --width 1280 --height 720 --batch 1024 --steps 20" 2>&1 | tee -a gpurun_out/r04p_k3_tail_ab.txt
echo "--- head tile width" | tee -a gpurun_out/r04p_k3_tail_ab.txt
for tc in 14 10 7; do run $ROOT new "" BSX_SEG_TILES=4,$tc,4,7,16,14,16,14; done 2>&1 | tee -a gpurun_out/r04p_k3_tail_ab.txt
for tc in 13 11 8; do run $ROOT new "--model mlkit --width 1280 --height 720 --steps 40" BSX_SEG_TILES=4,$tc,4,7,16,14,16,14; done 2>&1 | tee -a gpurun_out/r04p_k3_tail_ab.txt
