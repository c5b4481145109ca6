"""Print the per-micro-op timeline of the per-frame network program (run on a GPU box).
Usage: python tools/program_timeline.py [lite|full|mlkit|deeplab] [n_streams] [W H]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import backscrub_amd  # noqa: E402
from backscrub_amd import synth  # noqa: E402
from conftest import MODEL_KEYS, model_path  # noqa: E402

key = sys.argv[1] if len(sys.argv) > 1 else "lite"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 and sys.argv[3].isdigit() else (640, 480)
mg = backscrub_amd.MaskGen(model_path(key) if key in MODEL_KEYS else key, W, H, n_streams=n)
frames = torch.from_numpy(synth.frames(n, W, H, distinct=4)).cuda()
mg.run_stage(0, frames)
for _ in range(3):
    tl = mg.program_timeline(n)
steps = [l for l in mg.plan().splitlines() if l.startswith("P") and l[1:2].isdigit()]
tot = sum(tl)
for i, us in enumerate(tl):
    print("%6.2f us  %5.1f%%  %s" % (us, 100 * us / tot, steps[i] if i < len(steps) else "?"))
if "--fine" in sys.argv:
    print("fine (shader cycles): per op  [wave0: wait | dma | body]   max-body over waves   min-body   (waves with body > 500 cyc)")
    for i, row in enumerate(mg.last_fine):
        bodies = [r[2] for r in row]
        busy = sum(1 for b in bodies if b > 500)
        print("  P%-2d wait %6d dma %5d body %6d | body max %6d min %6d busy waves %2d | waitmax %6d" % (i, row[0][0], row[0][1], row[0][2], max(bodies), min(bodies), busy, max(r[0] for r in row)))
if "--fine" in sys.argv:
    ph = mg.last_subphase_us
    print("SE ops, cycles summed over all of them (wave 0): preload %d | pool %d | fc1 %d | barrier %d | fc2 %d" % tuple(int(ph[i] * 100) for i in (4, 5, 6, 7, 8)))
print("sub-phase accumulators (us):", [round(v, 1) for v in mg.last_subphase_us])
print("total %.1f us for workgroup 0 (n=%d)" % (tot, n))
print([l for l in mg.plan().splitlines() if l.startswith("frame program")][0])
