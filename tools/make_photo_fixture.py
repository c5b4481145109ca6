"""Generate tests/golden/photo_2x640x480.png + photo_expect.json — a REAL-PHOTO fixture for the end-to-end parity tests.

Source: the reference's own demo picture /root/reference/backgrounds/screenshot.jpg (1280x480 = two 640x480 webcam
screenshots of a person in front of two virtual backgrounds).  The JPEG is decoded once, here, with PIL and stored
losslessly as PNG so that every box sees identical pixels (JPEG decoders differ by an LSB between library versions).
photo_expect.json records what the CPU oracle makes of the two frames with the reference's four real models
(foreground fraction of the 3rd-frame mask + mask digests): DeepLab does not fire on the synthetic figure of
backscrub_amd/synth.py, so this photo is what gives its end-to-end test a non-empty person region.

    python tools/make_photo_fixture.py      # needs /root/reference (this container only); output is committed
"""
import hashlib
import json
import os
import sys

import numpy as np
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
SRC = "/root/reference/backgrounds/screenshot.jpg"
PNG = os.path.join(ROOT, "tests", "golden", "photo_2x640x480.png")
EXP = os.path.join(ROOT, "tests", "golden", "photo_expect.json")


def load_frames():
    """→ [2, 480, 640, 3] u8 BGR (the byte order cv::VideoCapture hands to the reference)"""
    rgb = np.asarray(Image.open(PNG).convert("RGB"))
    assert rgb.shape == (480, 1280, 3)
    return np.ascontiguousarray(np.stack([rgb[:, :640, ::-1], rgb[:, 640:, ::-1]]))


if __name__ == "__main__":
    from conftest import MODEL_KEYS, model_path
    from oracle import oracle_py as O
    Image.open(SRC).convert("RGB").save(PNG, optimize=True)
    frames = load_frames()
    out = {"source": "backgrounds/screenshot.jpg of floe/backscrub, PIL %s decode" % Image.__version__, "frames_sha256": hashlib.sha256(frames.tobytes()).hexdigest()}
    for key in MODEL_KEYS:
        rec = []
        for i in range(2):
            ctx = O.Ctx(model_path(key), 640, 480)
            for _ in range(3):
                m = ctx.process(frames[i])
            rec.append({"fg_fraction": round(float((m < 128).mean()), 5), "mask_sha256": hashlib.sha256(m.tobytes()).hexdigest()[:24]})
            ctx.close()
        out[key] = rec
        print(key, rec)
    json.dump(out, open(EXP, "w"), indent=1, sort_keys=True)
