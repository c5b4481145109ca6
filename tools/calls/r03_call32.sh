#!/bin/bash
# Gaussian blur tile height 16 / 32 / 64: bit-exactness tests under each, then timing at ksize 25 and 5 on 256 VGA and HD frames
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out
for th in 16 32 64; do BSX_GAUSS_TH=$th timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "gaussian or blur_own" 2>&1 | tail -1; done
cat > /tmp/g.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import backscrub_amd
from conftest import synthetic_model_path
for (W, H, n) in ((640, 480, 256), (1280, 720, 256)):
    mg = backscrub_amd.MaskGen(synthetic_model_path("lite"), W, H, n_streams=1)
    src = torch.randint(0, 256, (n, H, W, 3), dtype=torch.uint8, device="cuda")
    for k in (25, 5):
        for _ in range(3): out = mg.gaussian_blur(src, k)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(20): out = mg.gaussian_blur(src, k)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
        print(os.environ.get("BSX_GAUSS_TH"), "%dx%d x%d ksize %d: %.3f ms" % (W, H, n, k, dt * 1e3))
    mg.close()
PY
for th in 16 32 64; do BSX_GAUSS_TH=$th python /tmp/g.py 2>/dev/null; done
