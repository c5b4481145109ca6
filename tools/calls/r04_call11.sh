#!/bin/bash
# round 4, call 11: bsx_step_batch_pipelined — parity (bit-identical one call later, 6 geometries / flag sets, default + side stream), then the default bench line
# (every config carries a `pipelined` figure beside its synchronous `value`), then configs[1] again with the composite stream at default priority
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_batch.py -x -q -m gpu -k "pipelined" 2>&1 | tail -15 | tee gpurun_out/r04j_pytest.txt
grep -q "failed\|error" gpurun_out/r04j_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
timeout 600 python bench.py > gpurun_out/r04j_bench.json 2> gpurun_out/r04j_bench.err; tail -c 400 gpurun_out/r04j_bench.err
python - <<'P'
import json
d = json.load(open("gpurun_out/r04j_bench.json"))
print("configs[1]", d["value"], d["ms_per_step"], json.dumps(d.get("pipelined")))
for c in d.get("configs", []):
    print(c["baseline_config"], c["value"], c["ms_per_step"], json.dumps(c.get("pipelined")))
print(json.dumps(d["roofline_blend"])[:700])
P
BSX_PIPE_PRIO=0 timeout 300 python bench.py --no-extra-configs --no-cpu-baseline > gpurun_out/r04j_bench_prio0.json 2>/dev/null
python - <<'P'
import json
d = json.load(open("gpurun_out/r04j_bench_prio0.json"))
print("BSX_PIPE_PRIO=0 configs[1]", d["value"], d["ms_per_step"], json.dumps(d.get("pipelined")))
P
