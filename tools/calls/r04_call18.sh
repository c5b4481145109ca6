#!/bin/bash
# round 4, call 18: k3 / tail tile heights now that their LDS footprints changed (MLKit k3 at 16 rows = 45 KB = 3 workgroups per CU; 11 rows = 4)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
run() { cd $1; env $4 timeout 200 python bench.py --no-cpu-baseline --no-host-io --no-extra-configs --profile-iters 3 --steps 60 --warmup 10 --ramp-seconds 0.5 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}
print('$3', '$4', 'step', d['ms_per_step'], 'fps', d['value'], 'k2', t.get('seg_k2'), 'k3', t.get('seg_k3'), 'tail', t.get('seg_tail+decode'))"; }
M="--model mlkit --width 1280 --height 720 --steps 40"
for i in 1 2; do
run $ROOT new "$M" X=1
run $ROOT new "$M" BSX_SEG_TILES=4,13,4,7,12,13,16,13
run $ROOT new "$M" BSX_SEG_TILES=4,13,4,7,8,13,16,13
run $ROOT new "$M" BSX_SEG_TILES=4,13,4,7,16,13,12,13
run $ROOT new "$M" BSX_SEG_TILES=4,13,4,7,16,13,18,13
done 2>&1 | tee gpurun_out/r04q_k3_tail_tiles.txt
F="--model full --width 1280 --height 720 --batch 1024 --steps 20"
run $ROOT new "$F" X=1 2>&1 | tee -a gpurun_out/r04q_k3_tail_tiles.txt
run $ROOT new "$F" BSX_SEG_TILES=4,13,4,7,12,13,18,13 2>&1 | tee -a gpurun_out/r04q_k3_tail_tiles.txt
run $ROOT new "$F" BSX_SEG_TILES=4,13,4,7,9,13,15,13 2>&1 | tee -a gpurun_out/r04q_k3_tail_tiles.txt
