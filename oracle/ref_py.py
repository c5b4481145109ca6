"""ctypes binding of oracle/_ref/libbs_ref.so — the REFERENCE'S OWN in-tree sources (lib/libbackscrub.cc,
lib/transpose_conv_bias.cc, app/deepseg.cc:87-134) compiled unmodified against the API shims of oracle/ref_shim/.

TEST INFRASTRUCTURE ONLY.  Used by tests/test_ref_pin.py to pin the oracle's restatement of every piece of in-tree
reference code to the reference's object code.  The library is built by `make -C oracle ref-lib` where /root/reference
exists (this container); on the GPU box the prebuilt .so travels with the snapshot."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(_HERE, "_ref", "libbs_ref.so")
REF = os.environ.get("BSX_REFERENCE", "/root/reference")
_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
_i32p = C.POINTER(C.c_int)
DEBUG_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_char_p)
STAGE_FN = C.CFUNCTYPE(None, C.c_void_p)
_lib = None


def build(quiet=True):
    """(Re)build where the reference checkout exists; a no-op elsewhere."""
    if os.path.isdir(os.path.join(REF, "lib")):
        subprocess.check_call(["make", "-C", _HERE, "ref-lib", "REF=" + REF], stdout=subprocess.DEVNULL if quiet else None)


def available() -> bool:
    return os.path.exists(PATH) or os.path.isdir(os.path.join(REF, "lib"))


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(PATH)
        L.ref_tensorflow_version.restype = C.c_char_p
        L.ref_maskgen_new.restype = C.c_void_p
        L.ref_maskgen_new.argtypes = [C.c_char_p, C.c_long, C.c_long, C.c_long, DEBUG_FN, STAGE_FN, STAGE_FN, STAGE_FN, C.c_void_p]
        L.ref_maskgen_delete.argtypes = [C.c_void_p]
        L.ref_maskgen_process.argtypes = [C.c_void_p, _u8p, C.c_int, C.c_int, _u8p, C.POINTER(C.c_void_p)]
        L.ref_last_ofinal.argtypes = [_u8p, C.c_long, _i32p, _i32p]
        L.ref_force_output.argtypes = [_f32p, C.c_long]
        L.ref_alpha_blend.argtypes = [_u8p, _u8p, _u8p, _u8p, C.c_int, C.c_int]
        L.ref_convert_rgb_to_yuyv.argtypes = [_u8p, C.c_int, C.c_int, _u8p]
        L.ref_tconv_bias.argtypes = [_f32p, _i32p, _f32p, _i32p, _f32p, C.c_int, C.c_int, C.c_int, _f32p, _i32p]
        _lib = L
    return _lib


def _u8(a):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u8p)


def _f32(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f32p)


class RefMaskGen:
    """bs_maskgen_new / _process / _delete of the reference (lib/libbackscrub.cc:161-376), its own object code."""

    def __init__(self, model_path, width, height, threads=2, ondebug=None, onprep=None, oninfer=None, onmask=None):
        self._cbs = (DEBUG_FN(ondebug) if ondebug else DEBUG_FN(), STAGE_FN(onprep) if onprep else STAGE_FN(),
                     STAGE_FN(oninfer) if oninfer else STAGE_FN(), STAGE_FN(onmask) if onmask else STAGE_FN())
        self.h = lib().ref_maskgen_new(os.fsencode(model_path), threads, width, height, *self._cbs, None)
        self.width, self.height = width, height
        self.last_mask_ptr = None
        self._forced = None

    def ok(self):
        return bool(self.h)

    def process(self, frame: np.ndarray, forced_output: np.ndarray | None = None):
        """→ mask [H,W] u8, or None when the reference returns false.  forced_output: logits handed to the reference's
        decode in place of the network result."""
        frame = np.ascontiguousarray(frame)
        mask = np.empty((self.height, self.width), np.uint8)
        if forced_output is not None:
            self._forced = np.ascontiguousarray(forced_output, np.float32)
            lib().ref_force_output(_f32(self._forced), self._forced.size)
        p = C.c_void_p()
        try:
            rc = lib().ref_maskgen_process(self.h, _u8(frame), self.width, self.height, _u8(mask), C.byref(p))
        finally:
            lib().ref_force_output(None, 0)
        self.last_mask_ptr = p.value
        return mask if rc else None

    def last_ofinal(self):
        w, h = C.c_int(), C.c_int()
        n = lib().ref_last_ofinal(None, 0, C.byref(w), C.byref(h))
        out = np.empty((h.value, w.value), np.uint8)
        lib().ref_last_ofinal(_u8(out), n, None, None)
        return out

    def close(self):
        if self.h:
            lib().ref_maskgen_delete(self.h)
        self.h = None


def alpha_blend(bg, frame, mask):
    bg, frame, mask = (np.ascontiguousarray(x) for x in (bg, frame, mask))
    out = np.empty_like(frame)
    lib().ref_alpha_blend(_u8(bg), _u8(frame), _u8(mask), _u8(out), mask.shape[1], mask.shape[0])
    return out


def convert_rgb_to_yuyv(img):
    img = np.ascontiguousarray(img)
    out = np.zeros((img.shape[0], img.shape[1], 2), np.uint8)
    lib().ref_convert_rgb_to_yuyv(_u8(img), img.shape[1], img.shape[0], _u8(out))
    return out


def tconv_bias(x, w, b, padding=1, stride=(2, 2)):
    """x [1,H,W,Ci], w [Co,kh,kw,Ci], b [Co] → y [1,OH,OW,Co] through the reference's registered Prepare + Eval."""
    x, w, b = (np.ascontiguousarray(a, np.float32) for a in (x, w, b))
    xs, ws, ys = (C.c_int * 4)(*x.shape), (C.c_int * 4)(*w.shape), (C.c_int * 4)()
    if lib().ref_tconv_bias(_f32(x), xs, _f32(w), ws, _f32(b), padding, stride[0], stride[1], None, ys) != 0:
        raise RuntimeError("reference Convolution2DTransposeBias failed")
    y = np.empty(tuple(ys), np.float32)
    lib().ref_tconv_bias(_f32(x), xs, _f32(w), ws, _f32(b), padding, stride[0], stride[1], _f32(y), ys)
    return y
