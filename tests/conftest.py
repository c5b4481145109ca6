import glob
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _real_models():
    return sorted(glob.glob(os.path.join(ROOT, "models", "*.tflite")))


MODEL_KEYS = {
    "lite": "segm_lite_v681.tflite",
    "full": "segm_full_v679.tflite",
    "mlkit": "selfiesegmentation_mlkit-256x256-2021_01_19-v1215.f16.tflite",
    "deeplab": "deeplabv3_257_mv_gpu.tflite",
}


def model_path(key, prefer_real=True):
    """Path of a model for tests: the reference's real .tflite when it was staged into
    models/ (tools/stage_models.py; git-ignored, travels to the GPU box), else the synthetic
    same-architecture model with seeded random weights under tests/golden/models."""
    real = os.path.join(ROOT, "models", MODEL_KEYS[key])
    if prefer_real and os.path.exists(real):
        return real
    return synthetic_model_path(key)


def synthetic_model_path(key):
    from tools import make_synthetic_model
    return make_synthetic_model.ensure(key)


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py


@pytest.fixture
def debug_switches(monkeypatch):
    """The A/B knobs, alternate code paths and work-skipping experiments (csrc/debug_switches.hpp) are compiled out of libbsx.so; tests that cross-check an
    alternate path against the default one run against libbsx_dbg.so (same sources, -DBSX_DEBUG_SWITCHES; built by backscrub_amd/build.py).  For the duration of the
    test the Python binding is re-pointed at it; contexts must be closed inside the test."""
    from backscrub_amd import api, build
    if not os.path.exists(build.LIB_DBG):
        pytest.fail("libbsx_dbg.so is missing: run `python -m backscrub_amd.build`")
    saved = api._LIB
    api._LIB = None
    monkeypatch.setenv("BSX_LIBRARY", build.LIB_DBG)
    api.lib()
    yield api
    api._LIB = saved
