#!/bin/bash
# round 4, call 4: lane-coalesced compositor (blend4x4_k; dwordx3 stores in the fused tile kernel) — parity, then same-box A/B against _ab_old (previous commit)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mask or blend or step or composite or yuyv or flips or twin or in_place" 2>&1 | tail -3 | tee gpurun_out/r04_call4_pytest.txt
grep -q "failed\|error" gpurun_out/r04_call4_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
run() { cd $1; timeout 200 python bench.py --no-cpu-baseline --no-host-io --profile-iters 3 --steps 100 --warmup 10 --ramp-seconds 0.5 $3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}
print('$2', '$3', 'step', d['ms_per_step'], 'mask_blend', t.get('mask_blend'), 'blend_alone', d['roofline_blend']['avg_ms'], 'per_stream_bg', (d.get('roofline_blend_per_stream_bg') or {}).get('avg_ms'), 'yuyv', (d.get('yuyv_out') or {}).get('ms_per_step'))"; }
for i in 1 2; do run $ROOT/_ab_old old "--no-extra-configs"; run $ROOT new "--no-extra-configs"; done 2>&1 | tee gpurun_out/r04_call4_ab.txt
# the legs that only the default job prints (stand-alone blend with one background per stream, YUYV)
run2() { cd $1; timeout 300 python bench.py --no-cpu-baseline --no-host-io --profile-iters 3 --steps 60 --warmup 10 --ramp-seconds 0.5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$2', 'step', d['ms_per_step'], 'blend_alone', d['roofline_blend']['avg_ms'], 'per_stream_bg', d['roofline_blend_per_stream_bg']['avg_ms'], d['roofline_blend_per_stream_bg']['frac'], 'yuyv', d['yuyv_out']['ms_per_step'], 'bgblur', d['bgblur_step']['ms_per_step'], [(c['baseline_config'], c.get('ms_per_step')) for c in d['configs']])"; }
run2 $ROOT/_ab_old old 2>&1 | tee -a gpurun_out/r04_call4_ab.txt
run2 $ROOT new 2>&1 | tee -a gpurun_out/r04_call4_ab.txt
