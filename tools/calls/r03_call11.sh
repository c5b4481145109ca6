#!/bin/bash
# validation of HEAD after the container was re-created: GPU tests, the default bench line, kernel stats of the default job
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
T0=$SECONDS
timeout 1200 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/r03g_pytest.txt 2>&1
tail -14 gpurun_out/r03g_pytest.txt; echo "pytest $((SECONDS-T0)) s"; T0=$SECONDS
timeout 900 python bench.py > gpurun_out/r03g_bench.json 2> gpurun_out/r03g_bench.err
echo "bench $((SECONDS-T0)) s"; tail -3 gpurun_out/r03g_bench.err
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03g_bench.json'))
print(d['value'], d['ms_per_step'], d.get('roofline'))
print([(t['name'], t['ms']) for t in d['top_launches']])
for c in d.get('configs', []):
    print(c.get('workload'), c.get('value'), c.get('ms_per_step'), c.get('parity_sample'))
print(d.get('cpu_baseline'))
PY
