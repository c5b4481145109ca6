#!/bin/bash
# round 4, call 27 (evidence only): the default bench line of the final tree once more, on whatever box the pool hands out this time (the pool's boxes differ by a few per cent)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/r04z2_bench_other_box.json 2> /dev/null
python - <<'P'
import json
d = json.load(open('gpurun_out/r04z2_bench_other_box.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic_stale'], d['roofline']['frac_counted_traffic'])
print([(c['baseline_config'], c['value'], c['ms_per_step']) for c in d['configs']])
print([(c.get('BSX_F16_GEMM'), c['value']) for c in d['gemm_modes']], d['pipelined']['value'], d['yuyv_out']['value'])
P
