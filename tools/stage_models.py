"""Stage the reference's .tflite model DATA files into <repo>/models/ (git-ignored; travels to the GPU box with the snapshot).

The model file is what the user hands to bs_maskgen_new(modelname, ...) — data, not source.  Where /root/reference is absent
(the GPU box) whatever was staged earlier is used; tests fall back to tools/make_synthetic_model.py's same-architecture files.
"""
import glob
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("BSX_REFERENCE", "/root/reference")


def stage():
    dst = os.path.join(ROOT, "models")
    src = sorted(glob.glob(os.path.join(REF, "models", "*.tflite")))
    if not src:
        return dst
    os.makedirs(dst, exist_ok=True)
    for f in src:
        t = os.path.join(dst, os.path.basename(f))
        if not os.path.exists(t) or os.path.getmtime(t) < os.path.getmtime(f) or os.path.getsize(t) != os.path.getsize(f):
            shutil.copy2(f, t)
    # the reference's animated background (a DATA file: what `-b backgrounds/animated.gif` hands to load_background, app/background.cc:126-176) next to the models, so
    # that bench.py can time BASELINE configs[3] through the product's own background source on the GPU box
    bsrc = os.path.join(REF, "backgrounds", "animated.gif")
    if os.path.exists(bsrc):
        bdst = os.path.join(dst, "backgrounds")
        os.makedirs(bdst, exist_ok=True)
        t = os.path.join(bdst, "animated.gif")
        if not os.path.exists(t) or os.path.getsize(t) != os.path.getsize(bsrc):
            shutil.copy2(bsrc, t)
    return dst


if __name__ == "__main__":
    print(stage())
