#!/bin/bash
# round 4, call 25: prep_fused_k with one 8-byte load per source row — stage-0 parity (bit-exact), same-box A/B against _ab_old; the evidence is refreshed in the same call only if
# the new build is not slower (so that the committed profile matches the committed kernels either way: a slower build is reverted by the caller and the r04z files stay)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "stage or prep or stems or end_to_end" 2>&1 | tail -3 | tee gpurun_out/r04w_pytest.txt
grep -q "failed\|error" gpurun_out/r04w_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
run() { cd $1; timeout 200 python bench.py --no-cpu-baseline --no-host-io --no-extra-configs --no-side-probes --profile-iters 3 --steps 100 --warmup 10 --ramp-seconds 0.5 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}
print('$2', d['ms_per_step'], t.get('prep'), '$3')"; }
F="--model full --width 1280 --height 720 --batch 1024 --steps 20"
( for i in 1 2 3; do run $ROOT/_ab_old old ""; run $ROOT new ""; done; run $ROOT/_ab_old old "$F"; run $ROOT new "$F"; run $ROOT/_ab_old old "$F"; run $ROOT new "$F" ) 2>&1 | tee gpurun_out/r04w_prep_load8_ab.txt
OK=$(python - <<'PY'
rows = [l.split() for l in open('gpurun_out/r04w_prep_load8_ab.txt') if l.strip()]
def mean(tag, full): 
    v = [float(r[2]) for r in rows if r[0] == tag and (len(r) > 3) == full]
    return sum(v) / len(v)
print(1 if mean('new', False) <= mean('old', False) * 1.002 and mean('new', True) <= mean('old', True) * 1.002 else 0)
PY
)
echo "not_slower=$OK"
if [ "$OK" = "1" ]; then cd $ROOT; bash tools/calls/r04_call19.sh; fi
