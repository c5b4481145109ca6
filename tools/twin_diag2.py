"""Twin-stream determinism of the whole step at full batch (run on a GPU box): python tools/twin_diag2.py [key] [n]
16 scenes (at 640x480 the first two are the photo fixture) repeated through the batch; after each step ofinal / masks / composite of
every stream are compared with its scene twin; the logits too when BSX_KEEP_LOGITS=1."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import backscrub_amd  # noqa: E402
from backscrub_amd import synth  # noqa: E402
from conftest import model_path  # noqa: E402
from tools import make_photo_fixture  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "deeplab"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
W, H = 640, 480
distinct = 16
host = synth.frames(distinct, W, H)
host[:2] = make_photo_fixture.load_frames()
d = torch.from_numpy(host).cuda().repeat(n // distinct, 1, 1, 1).contiguous()
bg = torch.from_numpy(synth.background(W, H)).cuda()
out = torch.empty_like(d)
mg = backscrub_amd.MaskGen(model_path(key), W, H, n_streams=n)
print("env:", {k: v for k, v in os.environ.items() if k.startswith("BSX_")})


def twins(name, t):
    v = t.reshape(n // distinct, distinct, -1)
    bad = (v != v[0:1]).any(-1)
    cnt = (v != v[0:1]).sum(-1)
    where = [(int(i) * distinct + int(j), int(cnt[i, j])) for i, j in bad.nonzero()[:10].tolist()]
    print("  %-8s streams differing from their twin: %4d of %d   (stream, #values) %s" % (name, int(bad.sum()), n, where))
    return bad


for step in range(4):
    mg.step(d, bg, out)
    torch.cuda.synchronize()
    print("step", step)
    if os.environ.get("BSX_KEEP_LOGITS"):
        twins("logits", mg.output_tensor())
    twins("ofinal", mg.ofinal())
    twins("masks", mg.masks())
    twins("out", out)
