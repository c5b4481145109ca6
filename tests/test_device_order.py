"""The code path with `device != 0` on a box without a GPU (VERDICT r3, next #1b): libbsx.so's real host code — plan, hipRTC cache load, per-device kernel
attributes, hipGraph capture of the host path, every entry point's device guard — runs against tests/hip_stub/libhipstub.so, an LD_PRELOAD interposer that
pretends to have two devices and logs every HIP call with the calling thread's current device.  Asserted for a context on device 1 with the caller on device 0:
  * no device-affine HIP call (allocation, copy, launch, stream / event / module / graph operation, kernel attribute) is made while device 0 is current,
  * no handle is used on a device other than the one it was created on,
  * every entry point returns with the caller's device restored.
The hardware run of the same path is `bench.py --gpus N` (second_device_check) and tests/test_gpu_batch.py::test_context_on_a_second_device."""
import ctypes as C
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT, model_path

STUB_DIR = os.path.join(ROOT, "tests", "hip_stub")
STUB = os.path.join(STUB_DIR, "libhipstub.so")


@pytest.fixture(scope="module")
def stub():
    from backscrub_amd import build
    build.build()
    src = os.path.join(STUB_DIR, "hip_stub.cpp")
    if not os.path.exists(STUB) or os.path.getmtime(STUB) < os.path.getmtime(src):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", STUB, src])
    return STUB


def _drive(stub, tmp_path, key, W, H, n, dev, extra_env=None):
    log = str(tmp_path / ("hip_%s_%d.log" % (key, dev)))
    env = dict(os.environ, LD_PRELOAD=stub, BSX_STUB_LOG=log, BSX_STUB_NDEV="2")
    env.update(extra_env or {})
    r = subprocess.run([sys.executable, os.path.join(STUB_DIR, "drive.py"), model_path(key), str(W), str(H), str(n), str(dev)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    lines = [l.split() for l in open(log).read().splitlines() if l.strip()]
    return d, lines


@pytest.mark.parametrize("key,res", [("lite", (640, 480)), ("mlkit", (1280, 720)), ("deeplab", (640, 480))])
def test_context_on_device_1_never_touches_device_0(stub, tmp_path, key, res):
    d, lines = _drive(stub, tmp_path, key, res[0], res[1], 4, 1)
    assert "error" not in d, d
    assert d["device_in_info"] == 1
    bad_rc = {k: v for k, v in d["rc"].items() if v != 0}
    assert not bad_rc, bad_rc
    wrong = [(what, dev) for what, dev in d["caller_device"] if dev != 0]
    assert not wrong, "the caller's current device was not restored after: %s" % wrong
    affine = [l for l in lines if l[0] == "affine"]
    assert len(affine) > 100                                             # the run really went through the interposer
    apis = {l[1] for l in affine}
    for must in ("hipMalloc", "hipLaunchKernel", "hipFuncSetAttribute", "hipStreamCreateWithFlags", "hipStreamCreateWithPriority", "hipStreamWaitEvent", "hipGraphLaunch", "hipStreamBeginCapture", "hipEventRecord", "hipMemcpy2DAsync"):
        assert must in apis, "%s never reached: the driver does not cover that path" % must
    if key != "deeplab":
        assert d["specialised"] and "hipModuleLoadData" in apis and "hipModuleLaunchKernel" in apis       # the hipRTC kernel, loaded and launched on device 1
    off_device = [l for l in affine if int(l[2]) != 1]
    assert not off_device, "HIP calls made while device 0 was current: %s" % sorted({l[1] for l in off_device})
    assert not [l for l in lines if l[0] == "MISMATCH"], [l for l in lines if l[0] == "MISMATCH"][:5]
    queried = {l[3] for l in lines if l[1] == "hipGetDeviceProperties"}
    assert queried <= {"queried=1"}                                      # the architecture the kernel is compiled for is device 1's


def test_context_on_device_0_is_the_same_sequence(stub, tmp_path):
    """device 0 and device 1 contexts issue the same HIP call sequence (only the device differs): nothing is special-cased on the default device"""
    _, l0 = _drive(stub, tmp_path, "lite", 640, 480, 4, 0)
    _, l1 = _drive(stub, tmp_path, "lite", 640, 480, 4, 1)
    seq0 = [l[1] for l in l0 if l[0] == "affine"]
    seq1 = [l[1] for l in l1 if l[0] == "affine"]
    assert seq0 == seq1
    assert all(int(l[2]) == 0 for l in l0 if l[0] == "affine")


def test_the_interposer_catches_a_violation(stub, tmp_path):
    """negative control of the checker itself: a stream created on device 0 and used on device 1 is reported"""
    log = str(tmp_path / "neg.log")
    code = ("import ctypes as C, os\n"
            "s = C.CDLL(%r)\n"
            "h = C.c_void_p()\n"
            "s.hipStreamCreateWithFlags(C.byref(h), 0)\n"
            "s.hipSetDevice(1)\n"
            "s.hipStreamSynchronize(h)\n" % stub)
    subprocess.check_call([sys.executable, "-c", code], env=dict(os.environ, BSX_STUB_LOG=log))
    txt = open(log).read()
    assert "MISMATCH hipStreamSynchronize 1 owner=0" in txt
