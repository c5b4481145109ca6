#!/bin/bash
# round 6, call 4: BSX_STEP_YUYV_IN — parity of the fused YUYV-in step (six geometries, pipelined), the release / debug switch test, then the bench line with the yuyv_in_out and host_io_yuyv legs
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "yuyv or release_library or flips or pipelined" 2>&1 | tail -15 | tee gpurun_out/r06d_pytest.txt
grep -q "failed\|error" gpurun_out/r06d_pytest.txt && { echo "PARITY FAILED — stopping"; exit 1; }
timeout 900 python bench.py --no-cpu-baseline --no-side-probes > /dev/null 2>&1   # (warm the box)
timeout 900 python bench.py --no-cpu-baseline --steps 100 --warmup 10 --detail gpurun_out/r06d_bench_detail.json > gpurun_out/r06d_bench.json 2> gpurun_out/r06d_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06d_bench_detail.json'))
for k in ('value','ms_per_step','host_io','host_io_yuyv','yuyv_out','yuyv_in_out'): print(k, json.dumps(d.get(k))[:400])
PY
