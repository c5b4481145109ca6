#!/bin/bash
# round 6, call 18: specialised-vs-AOT segment kernels bit-identity test, then the full default bench line on the current tree
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "specialised_segment or release_library or stages_match" 2>&1 | tail -5 | tee gpurun_out/r06q_pytest.txt
timeout 1500 python bench.py --detail gpurun_out/r06q_bench_detail.json > gpurun_out/r06q_bench.json 2> gpurun_out/r06q_bench.err; tail -2 gpurun_out/r06q_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06q_bench.json').read().strip().splitlines()[-1])
print(len(json.dumps(d)), d['value'], d['ms_per_step'], d['roofline'])
for c in d.get('configs',[]): print({k:c.get(k) for k in ('baseline_config','value','ms_per_step','kernel','frac','iou_min','max_abs','bg_identical','h2d_ring_value')})
for k in ('host_io','host_io_yuyv','yuyv_in_out','yuyv_out','worst_case','static_scene','stage_ms','top_launches'): print(k, d.get(k))
PY
