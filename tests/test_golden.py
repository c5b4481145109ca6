"""Regression digests of the CPU oracle (tests/golden/oracle_digests.json, written by tools/make_golden.py).
Self-generated — they freeze the oracle, they do not pin it to the reference (which ships no vectors)."""
import json
import os

from conftest import ROOT


def test_oracle_digests_unchanged(oracle):
    from tools import make_golden
    want = json.load(open(os.path.join(ROOT, "tests", "golden", "oracle_digests.json")))
    got = make_golden.compute()
    assert got == want, "the oracle (or synth.py / the synthetic model generator) changed: re-run tools/make_golden.py if intended"


def test_photo_fixture_person_is_segmented_by_all_four_models(oracle):
    """Semantic pin of the oracle with the reference's REAL weights on the reference's own demo photo (tests/golden/
    photo_2x640x480.png, made by tools/make_photo_fixture.py from backgrounds/screenshot.jpg): every one of the four networks
    finds the person — ~24 % of the frame — and the masks are the ones recorded when the fixture was made."""
    import hashlib

    import pytest

    from conftest import MODEL_KEYS, model_path
    from tools import make_photo_fixture as P
    frames = P.load_frames()
    want = json.load(open(P.EXP))
    assert hashlib.sha256(frames.tobytes()).hexdigest() == want["frames_sha256"]
    ran = 0
    for key in MODEL_KEYS:
        path = model_path(key)
        if "synthetic" in os.path.basename(path):
            continue                      # random weights segment nothing: only meaningful with the reference's model files
        ran += 1
        for i in (0, 1):
            ctx = oracle.Ctx(path, 640, 480)
            for _ in range(3):
                m = ctx.process(frames[i])
            ctx.close()
            fg = float((m < 128).mean())
            assert 0.20 <= fg <= 0.28, "%s frame %d: foreground %.4f" % (key, i, fg)
            assert abs(fg - want[key][i]["fg_fraction"]) <= 2e-3, "%s frame %d: %.5f vs recorded %.5f" % (key, i, fg, want[key][i]["fg_fraction"])
    if not ran:
        pytest.skip("reference model files not staged on this box")
