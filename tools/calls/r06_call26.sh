#!/bin/bash
# round 6, call 26: the default bench as the driver runs it, on the committed tree with its own counter passes (profiles/pmc_latest.json carries the tree's digest)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r06y_bench_driver_detail.json > gpurun_out/r06y_bench_driver.json 2> gpurun_out/r06y_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06y_bench_driver.json').read().strip().splitlines()[-1])
print(len(json.dumps(d)), d['value'], d['ms_per_step'], d['roofline'])
for k in ('static_scene','worst_case','yuyv_in_out','host_io','host_io_yuyv'): print(k, d.get(k))
for c in d.get('configs',[]): print({k:c.get(k) for k in ('baseline_config','value','ms_per_step','frac','kernel','kernel_ms','iou_min','max_abs','h2d_ring_value','bg_identical')})
PY
