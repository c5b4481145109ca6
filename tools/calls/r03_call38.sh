#!/bin/bash
# gauss_blur_k variants (word stores / 4-pixel staging), BSX_STEP_BGBLUR one pass vs two calls, seg_k2 weight prefetch
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "gaussian or own_blur or blur_own" 2>&1 | tail -4
cat > /tmp/gt.py <<'PY'
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import backscrub_amd as bs  # noqa
from tests.conftest import model_path
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(it): fn()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / it
for (W, H) in ((640, 480), (1280, 720)):
    n = 256
    mg = bs.MaskGen(model_path("lite"), W, H, n_streams=n)
    fr = torch.randint(0, 256, (n, H, W, 3), dtype=torch.uint8, device="cuda")
    o1, o2, bl = torch.empty_like(fr), torch.empty_like(fr), torch.empty_like(fr)
    line = ["%dx%d" % (W, H)]
    for k in (25, 5, 3):
        line.append("k%d %.3f" % (k, t(lambda: mg.gaussian_blur(fr, k, out=bl))))
    def two():
        mg.gaussian_blur(fr, 25, out=bl); mg.step(fr, bl, o1)
    def one():
        mg.step_ex(fr, None, o2, bgblur=25)
    line.append("step %.3f" % t(lambda: mg.step(fr, bl, o1)))
    line.append("two-call %.3f one-pass %.3f" % (t(two), t(one)))
    print(os.environ.get("TAG", ""), " | ".join(line), flush=True)
    mg.close()
PY
TAG="words+stage4 " timeout 200 python /tmp/gt.py 2>&1 | tail -2
TAG="bytes+stage4 " BSX_GAUSS_BYTE_STORE=1 timeout 200 python /tmp/gt.py 2>&1 | tail -2
TAG="words+bytestg" BSX_GAUSS_BYTE_STAGE=1 timeout 200 python /tmp/gt.py 2>&1 | tail -2
TAG="bytes+bytestg" BSX_GAUSS_BYTE_STORE=1 BSX_GAUSS_BYTE_STAGE=1 timeout 200 python /tmp/gt.py 2>&1 | tail -2
run() { env $1 timeout 200 python bench.py --no-extra-configs --no-cpu-baseline --profile-iters 3 --steps 60 --warmup 10 --ramp-seconds 0.5 $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t={x['name']:x['ms'] for x in d['top_launches']}; print('$1 $2', d['ms_per_step'], [t.get(k) for k in ('seg_head','seg_k2','seg_k3','seg_tail+decode')])"; }
run BSX_K2_PREFETCH=0
run BSX_K2_PREFETCH=1
run BSX_K2_PREFETCH=0
run BSX_K2_PREFETCH=1
run BSX_K2_PREFETCH=0 "--model mlkit --width 1280 --height 720"
run BSX_K2_PREFETCH=1 "--model mlkit --width 1280 --height 720"
