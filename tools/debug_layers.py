"""Layer-by-layer comparison of the GPU network against the CPU oracle (run on a GPU box).
Usage: BSX_ARENA_NO_REUSE=1 python tools/debug_layers.py <model.tflite|key> [W H]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("BSX_ARENA_NO_REUSE", "1")
import torch  # noqa: E402

import backscrub_amd  # noqa: E402
from backscrub_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402

sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import MODEL_KEYS, model_path  # noqa: E402

arg = sys.argv[1] if len(sys.argv) > 1 else "lite"
path = model_path(arg) if arg in MODEL_KEYS else arg
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480)
mg = backscrub_amd.MaskGen(path, W, H, n_streams=1)
print(mg.plan())
oc = O.Ctx(path, W, H)
f = synth.frame(W, H, 0)
mg.run_stage(0, torch.from_numpy(f[None]).cuda())
mg.run_stage(1, n=1)
torch.cuda.synchronize()
oc.prep(f)
oc.infer()
om = oc.model()
bad = 0
for t in range(om.n_tensors):
    try:
        g = mg.graph_tensor(t)
    except backscrub_amd.BsxError:
        continue
    w = om.tensor(t).ravel()
    if w.size != g.size or w.size == 0:
        continue
    scale = max(1.0, float(np.abs(w).max()))
    err = float(np.abs(g - w).max()) / scale
    flag = "" if err < 1e-4 else "  <<<<<< MISMATCH"
    if flag:
        bad += 1
    print("tensor %3d shape %-18s rel err %.3g%s" % (t, om.shape(t), err, flag))
print("mismatching tensors:", bad)
