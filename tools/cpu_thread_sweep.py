"""CPU baseline of the oracle port at 1, 2 and all host threads (the reference's --debug-timing thread counts).
Usage: python tools/cpu_thread_sweep.py [lite|full|mlkit|deeplab] [W H]   → one JSON line"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from backscrub_amd import synth  # noqa: E402
from conftest import model_path  # noqa: E402
from oracle import oracle_py  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "lite"
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480)
oracle_py.build()
mp = model_path(key)
bg = synth.background(W, H)
out = {"model": os.path.basename(mp), "frame": [W, H]}
for th in (1, 2, os.cpu_count()):
    frames = synth.frames(th, W, H, distinct=min(th, 4))
    oracle_py.baseline_run(mp, frames, bg, 1, th)
    sec, _, _ = oracle_py.baseline_run(mp, frames, bg, 2, th)
    iters = int(max(2, min(400, 6.0 / max(sec / 2, 1e-3))))
    sec, st, _ = oracle_py.baseline_run(mp, frames, bg, iters, th)
    out["threads_%d" % th] = {"fps": round(th * iters / sec, 1), "ms_per_frame_per_thread": round(1e3 * sec / iters, 3),
                              "stage_share": [round(s / (sum(st) or 1), 3) for s in st]}
print(json.dumps(out))
