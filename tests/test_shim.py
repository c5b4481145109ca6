"""The C++ drop-in face (bs_maskgen_* with the reference's signatures) exercised from a C++
application with no Python/torch in the process: tools/bsx_demo.cpp, built against tests/cv_stub."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, model_path


@pytest.fixture(scope="module")
def demo():
    from backscrub_amd import build
    build.build()
    return build.DEMO


def test_demo_reports_missing_model_like_the_reference(demo, tmp_path):
    r = subprocess.run([demo, str(tmp_path / "segm_missing.tflite"), "640", "480", "x", "1", "y"], capture_output=True, text=True)
    assert r.returncode == 3
    assert "unable to load model from file" in r.stderr      # lib/libbackscrub.cc:191-195 wording
    assert "gfx950" in r.stdout


@pytest.mark.gpu
def test_demo_masks_match_oracle(demo, oracle, tmp_path):
    from backscrub_amd import synth
    W, H, T = 640, 480, 4
    path = model_path("lite")
    frames = np.stack([synth.frame(W, H, 2, t) for t in range(T)])
    fin, fout = tmp_path / "frames.bgr", tmp_path / "masks.u8"
    frames.tofile(fin)
    r = subprocess.run([demo, path, str(W), str(H), str(fin), str(T), str(fout)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(fout, np.uint8).reshape(T, H, W)
    oc = oracle.Ctx(path, W, H)
    for t in range(T):
        want = oc.process(frames[t])
        fa, fb = got[t] < 128, want < 128
        union = np.logical_or(fa, fb).sum()
        assert union == 0 or np.logical_and(fa, fb).sum() / union >= 0.999, "frame %d" % t
    oc.close()
