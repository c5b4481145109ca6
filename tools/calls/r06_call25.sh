#!/bin/bash
# round 6, call 25: the tree with the chained ASPP head — full GPU suite, smoke, the default bench as the driver runs it, then the evidence set (r06z_*, pmc_latest.json)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/r06y_pytest.txt
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a gpurun_out/r06y_pytest.txt
timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 --detail gpurun_out/r06y_bench_driver_detail.json > gpurun_out/r06y_bench_driver.json 2> gpurun_out/r06y_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06y_bench_driver.json').read().strip().splitlines()[-1])
print(len(json.dumps(d)), d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('traffic_stale'))
for c in d.get('configs',[]): print({k:c.get(k) for k in ('baseline_config','value','frac','iou_min','max_abs')})
PY
bash tools/calls/r06_evidence.sh r06z 2>&1 | tail -25
