#!/bin/bash
# hipGraph replay of the single-frame launch chain (bsx_process_host): whole GPU suite, then single-stream latency A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r03t_pytest.txt 2>&1; tail -4 gpurun_out/r03t_pytest.txt
cat > /tmp/lat.py <<'PY'
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench
for key, calls in (("lite", 300), ("mlkit", 200), ("deeplab", 100)):
    print(os.environ.get("BSX_NO_GRAPH"), bench.single_stream_latency(key, 640, 480, calls))
PY
for rep in 1 2; do
python /tmp/lat.py 2>/dev/null
BSX_NO_GRAPH=1 python /tmp/lat.py 2>/dev/null
done
