#!/bin/bash
# mask_tile_k horizontal pass four columns per item: mask / blend parity, same-box A/B against _ab_old (previous commit); if the new build is faster, refresh the
# default job's rocprofv3 kernel stats + HBM PMC (tag r03z) in the same call so that the committed profile matches the kernel
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; timeout 200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mask or blend or step or composite or yuyv or flips or twin" 2>&1 | tail -3 | tee /tmp/pt.txt
grep -q "failed\|error" /tmp/pt.txt && { echo "PARITY FAILED — stopping"; exit 1; }
run() { cd $1; timeout 200 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 3 --steps 100 --warmup 10 --ramp-seconds 0.5 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}; print('$3 $2', d['ms_per_step'], t.get('mask_blend'))"; }
run $ROOT/_ab_old "" old | tee /tmp/o1.txt
run $ROOT "" new | tee /tmp/n1.txt
run $ROOT/_ab_old "" old | tee /tmp/o2.txt
run $ROOT "" new | tee /tmp/n2.txt
run $ROOT/_ab_old "--model mlkit --width 1280 --height 720" old
run $ROOT "--model mlkit --width 1280 --height 720" new
FASTER=$(python - <<'PY'
def v(f): return float(open(f).read().split()[1])
o = (v('/tmp/o1.txt') + v('/tmp/o2.txt')) / 2; n = (v('/tmp/n1.txt') + v('/tmp/n2.txt')) / 2
print(1 if n < 0.99 * o else 0)
PY
)
echo "faster=$FASTER"
if [ "$FASTER" = "1" ]; then cd $ROOT; mkdir -p gpurun_out; timeout 150 bash tools/profile_round.sh r03z > gpurun_out/r03z_profile_round.log 2>&1; head -12 gpurun_out/r03z_kernel_stats.md; fi
