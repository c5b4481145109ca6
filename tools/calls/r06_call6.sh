#!/bin/bash
# round 6, call 6: the full default bench line (configs[3] through bsx_background_load / _grab; YUYV legs), as the driver runs it
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 1500 python bench.py --detail gpurun_out/r06e_bench_detail.json > gpurun_out/r06e_bench.json 2> gpurun_out/r06e_bench.err; tail -3 gpurun_out/r06e_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06e_bench.json').read().strip().splitlines()[-1])
print(len(json.dumps(d)), d['value'], d['ms_per_step'], d['roofline'])
for c in d.get('configs',[]): print(c)
for k in ('host_io','host_io_yuyv','yuyv_in_out','worst_case','opt_in_modes'): print(k, d.get(k))
PY
