"""Generate tests/golden/oracle_digests.json — regression digests of the CPU oracle on seeded synthetic inputs.

These are NOT reference-derived pins (the reference ships none and cannot be built here; the oracle is "parity unpinned",
see DESIGN.md §2).  They freeze today's oracle so that an accidental change to oracle/bs_oracle.cpp, to the synthetic
model generator or to backscrub_amd/synth.py is caught by `pytest -m "not gpu"` on any box.

    python tools/make_golden.py        # rewrites the digest file
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from backscrub_amd import synth  # noqa: E402
from oracle import oracle_py as O  # noqa: E402
from tools import make_synthetic_model as S  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "oracle_digests.json")
CASES = [("lite", 640, 480), ("full", 1280, 720), ("mlkit", 640, 480), ("deeplab", 640, 480)]


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:24]


def compute():
    out = {}
    for key, w, h in CASES:
        ctx = O.Ctx(S.ensure(key), w, h)
        bg = synth.background(w, h)
        rec = {}
        for t in range(3):
            f = synth.frame(w, h, 1, t)
            m = ctx.process(f)
            rec["t%d" % t] = {"frame": digest(f), "mask": digest(m), "ofinal": digest(ctx.ofinal()),
                              "composite": digest(O.alpha_blend(bg, f, m)), "fg_fraction": round(float((m < 128).mean()), 6)}
        # logits are float: digest a coarse quantisation so that libm ulp differences between boxes do not matter
        rec["logits_q"] = digest(np.round(ctx.output() * 64).astype(np.int32))
        out["%s_%dx%d" % (key, w, h)] = rec
        ctx.close()
    img = synth.random_u8((37, 53, 3), 21)
    out["image_ops"] = {
        "resize_down": digest(O.resize_linear(img, 20, 11)), "resize_up": digest(O.resize_linear(img[..., 0].copy(), 130, 97)),
        "bilateral": digest(O.bilateral(img)), "blur5": digest(O.blur5(img[..., 1].copy())),
        "yuyv": digest(O.bgr_to_yuyv(synth.random_u8((8, 16, 3), 22))), "yuyv_to_bgr": digest(O.yuyv_to_bgr(synth.random_u8((8, 16, 2), 23))),
    }
    return out


if __name__ == "__main__":
    O.build()
    d = compute()
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(d, f, indent=1, sort_keys=True)
    print("wrote", OUT)
