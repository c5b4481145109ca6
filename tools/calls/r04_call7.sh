#!/bin/bash
# round 4, call 7: tile classes per configuration (why did MLKit not gain?) + a look at the model-resolution masks themselves
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT; mkdir -p gpurun_out
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_call7_tiles.txt
import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import backscrub_amd
from backscrub_amd import synth
from conftest import model_path
for key, (W, H) in (("lite", (640, 480)), ("mlkit", (1280, 720)), ("mlkit", (640, 480)), ("full", (1280, 720)), ("deeplab", (640, 480))):
    n = 16
    mg = backscrub_amd.MaskGen(model_path(key), W, H, n_streams=n)
    fr = torch.from_numpy(synth.frames(n, W, H, t=0)).cuda()
    bg = torch.from_numpy(synth.background(W, H)).cuda()
    out = torch.empty_like(fr)
    for _ in range(5):
        mg.step(fr, bg, out)
    torch.cuda.synchronize()
    of = mg.ofinal().cpu().numpy()
    vals, cnt = np.unique(of, return_counts=True)
    i = mg.info
    q = i["in_roi"]
    sub = of[:, q[1]:q[1] + q[3], q[0]:q[0] + q[2]]
    # isolated pixels: differ from all 4 neighbours' majority — speckle measure
    a = (sub == 0)
    trans = (a[:, 1:, :] != a[:, :-1, :]).sum() + (a[:, :, 1:] != a[:, :, :-1]).sum()
    print(key, W, H, mg.mask_tile_stats(n), "ofinal values", dict(zip(vals.tolist(), cnt.tolist())), "person frac %.3f" % a.mean(), "edge transitions per frame %.0f" % (trans / n), "in_roi", q, "roi", i["roi"])
    mg.close()
PY
