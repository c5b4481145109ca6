#!/bin/bash
# Gaussian blur with dword staging / dword stores: bit-exactness tests, timing
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gaussian or blur_own" 2>&1 | tail -2
cat > /tmp/g.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import backscrub_amd
from conftest import synthetic_model_path
for (W, H, n) in ((640, 480, 256), (1280, 720, 256)):
    mg = backscrub_amd.MaskGen(synthetic_model_path("lite"), W, H, n_streams=1)
    src = torch.randint(0, 256, (n, H, W, 3), dtype=torch.uint8, device="cuda")
    for k in (25, 5, 3):
        for _ in range(3): out = mg.gaussian_blur(src, k)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): out = mg.gaussian_blur(src, k)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
        print("%dx%d x%d ksize %d: %.3f ms" % (W, H, n, k, dt * 1e3))
    mg.close()
PY
python /tmp/g.py 2>/dev/null
