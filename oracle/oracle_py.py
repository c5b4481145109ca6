"""ctypes binding of the CPU oracle (oracle/libbs_oracle*.so).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (backscrub_amd) never imports this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)


def build(quiet=True):
    subprocess.check_call(["make", "-C", _HERE, "libbs_oracle.so", "libbs_oracle_fast.so", "ref-lib"],
                          stdout=subprocess.DEVNULL if quiet else None)


def _load(name):
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.bso_version.restype = C.c_char_p
    lib.bso_model_load.restype = C.c_void_p
    lib.bso_model_load.argtypes = [C.c_char_p]
    lib.bso_model_free.argtypes = [C.c_void_p]
    for f in ("bso_model_num_ops", "bso_model_num_tensors", "bso_model_input", "bso_model_output"):
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.bso_model_tensor_shape.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib.bso_model_tensor_data.argtypes = [C.c_void_p, C.c_int, _f32p, C.c_long]
    lib.bso_model_tensor_data.restype = C.c_long
    lib.bso_model_op.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    lib.bso_model_invoke.argtypes = [C.c_void_p, _f32p, _f32p]
    lib.bso_resize_linear_u8.argtypes = [_u8p, C.c_int, C.c_int, C.c_long, C.c_int, _u8p, C.c_int, C.c_int, C.c_long]
    lib.bso_bilateral_c3.argtypes = [_u8p, C.c_int, C.c_int, _u8p, C.c_int, C.c_double, C.c_double]
    lib.bso_blur5_u8.argtypes = [_u8p, C.c_int, C.c_int, C.c_long, _u8p, C.c_long]
    lib.bso_alpha_blend.argtypes = [_u8p, _u8p, _u8p, _u8p, C.c_long]
    lib.bso_bgr_to_yuyv.argtypes = [_u8p, C.c_int, C.c_int, _u8p]
    lib.bso_yuyv_to_bgr.argtypes = [_u8p, C.c_int, C.c_int, _u8p]
    lib.bso_decode_iir.argtypes = [C.c_int, _f32p, C.c_long, C.c_int, _u8p]
    lib.bso_convert_f32.argtypes = [_u8p, C.c_long, C.c_float, C.c_float, _f32p]
    lib.bso_gaussian_blur_c3.argtypes = [_u8p, C.c_int, C.c_int, C.c_int, _u8p]
    lib.bso_gaussian_coeffs.argtypes = [C.c_int, C.POINTER(C.c_uint16)]
    lib.bso_tconv_bias.argtypes = [_f32p, C.POINTER(C.c_int), _f32p, C.POINTER(C.c_int), _f32p, C.c_int, C.c_int, C.c_int, _f32p, C.POINTER(C.c_int)]
    lib.bso_ctx_new.restype = C.c_void_p
    lib.bso_ctx_new.argtypes = [C.c_char_p, C.c_int, C.c_int]
    lib.bso_ctx_delete.argtypes = [C.c_void_p]
    lib.bso_ctx_process.argtypes = [C.c_void_p, _u8p, _u8p]
    lib.bso_ctx_geometry.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    for f, rt in (("bso_ctx_input", _f32p), ("bso_ctx_output", _f32p), ("bso_ctx_ofinal", _u8p), ("bso_ctx_mask", _u8p),
                  ("bso_ctx_model", C.c_void_p)):
        getattr(lib, f).restype = rt
        getattr(lib, f).argtypes = [C.c_void_p]
    lib.bso_ctx_set_ofinal.argtypes = [C.c_void_p, _u8p]
    lib.bso_ctx_prep.argtypes = [C.c_void_p, _u8p]
    lib.bso_ctx_infer.argtypes = [C.c_void_p]
    lib.bso_ctx_set_output.argtypes = [C.c_void_p, _f32p]
    lib.bso_ctx_post.argtypes = [C.c_void_p]
    lib.bso_baseline_run.restype = C.c_double
    lib.bso_baseline_run.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _u8p, _u8p, _u8p,
                                     C.POINTER(C.c_double)]
    return lib


_libs = {}


def lib(fast=False):
    key = "libbs_oracle_fast.so" if fast else "libbs_oracle.so"
    if key not in _libs:
        _libs[key] = _load(key)
    return _libs[key]


def _u8(a):
    assert a.dtype == np.uint8 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u8p)


def _f32(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f32p)


# ---- image ops -------------------------------------------------------------------------
def resize_linear(src: np.ndarray, dw: int, dh: int) -> np.ndarray:
    src = np.ascontiguousarray(src)
    sh, sw = src.shape[:2]
    cn = 1 if src.ndim == 2 else src.shape[2]
    dst = np.empty((dh, dw) if src.ndim == 2 else (dh, dw, cn), np.uint8)
    lib().bso_resize_linear_u8(_u8(src), sw, sh, sw * cn, cn, _u8(dst), dw, dh, dw * cn)
    return dst


def bilateral(src: np.ndarray, d=5, sc=100.0, ss=100.0) -> np.ndarray:
    src = np.ascontiguousarray(src)
    dst = np.empty_like(src)
    lib().bso_bilateral_c3(_u8(src), src.shape[1], src.shape[0], _u8(dst), d, sc, ss)
    return dst


def blur5(src: np.ndarray) -> np.ndarray:
    src = np.ascontiguousarray(src)
    dst = np.empty_like(src)
    lib().bso_blur5_u8(_u8(src), src.shape[1], src.shape[0], src.shape[1], _u8(dst), src.shape[1])
    return dst


def alpha_blend(bg, fr, mask) -> np.ndarray:
    bg, fr, mask = (np.ascontiguousarray(x) for x in (bg, fr, mask))
    out = np.empty_like(fr)
    lib().bso_alpha_blend(_u8(bg), _u8(fr), _u8(mask), _u8(out), mask.size)
    return out


def bgr_to_yuyv(img) -> np.ndarray:
    img = np.ascontiguousarray(img)
    out = np.zeros((img.shape[0], img.shape[1], 2), np.uint8)
    lib().bso_bgr_to_yuyv(_u8(img), img.shape[1], img.shape[0], _u8(out))
    return out


def gaussian_blur(img: np.ndarray, ksize: int) -> np.ndarray:
    """cv::GaussianBlur(img, out, Size(ksize, ksize), 0) on packed 8-bit BGR (/root/reference/app/deepseg.cc:657-658)."""
    img = np.ascontiguousarray(img)
    out = np.empty_like(img)
    lib().bso_gaussian_blur_c3(_u8(img), img.shape[1], img.shape[0], int(ksize), _u8(out))
    return out


def gaussian_coeffs(ksize: int) -> np.ndarray:
    out = np.zeros(ksize, np.uint16)
    lib().bso_gaussian_coeffs(int(ksize), out.ctypes.data_as(C.POINTER(C.c_uint16)))
    return out


def flip_bgr(img: np.ndarray, code: int) -> np.ndarray:
    """cv::flip(img, out, code) as the app applies it to the composited frame (/root/reference/app/deepseg.cc:667-673):
    code 0 reverses the rows, code > 0 the columns, code < 0 both."""
    if code == 0:
        return np.ascontiguousarray(img[::-1])
    if code > 0:
        return np.ascontiguousarray(img[:, ::-1])
    return np.ascontiguousarray(img[::-1, ::-1])


def yuyv_to_bgr(img) -> np.ndarray:
    img = np.ascontiguousarray(img)
    out = np.zeros((img.shape[0], img.shape[1], 3), np.uint8)
    lib().bso_yuyv_to_bgr(_u8(img), img.shape[1], img.shape[0], _u8(out))
    return out


def tconv_bias(x, w, b, padding=1, stride=(2, 2)) -> np.ndarray:
    """Convolution2DTransposeBias restatement: x [1,H,W,Ci], w [Co,kh,kw,Ci], b [Co] → [1,OH,OW,Co]."""
    x, w, b = (np.ascontiguousarray(a, np.float32) for a in (x, w, b))
    xs, ws, ys = (C.c_int * 4)(*x.shape), (C.c_int * 4)(*w.shape), (C.c_int * 4)()
    lib().bso_tconv_bias(_f32(x), xs, _f32(w), ws, _f32(b), padding, stride[0], stride[1], None, ys)
    y = np.empty(tuple(ys), np.float32)
    lib().bso_tconv_bias(_f32(x), xs, _f32(w), ws, _f32(b), padding, stride[0], stride[1], _f32(y), ys)
    return y


def decode_iir(modeltype: int, logits: np.ndarray, ofinal: np.ndarray) -> np.ndarray:
    logits = np.ascontiguousarray(logits, np.float32)
    out = np.ascontiguousarray(ofinal).copy()
    nch = logits.shape[-1] if logits.ndim == 3 else 1
    lib().bso_decode_iir(modeltype, _f32(logits), out.size, nch, _u8(out))
    return out


# ---- model ------------------------------------------------------------------------------
class Model:
    def __init__(self, path, handle=None, fast=False):
        self.L = lib(fast)
        self.owned = handle is None
        self.h = handle or self.L.bso_model_load(path.encode())
        if not self.h:
            raise RuntimeError("oracle: cannot load model %s" % path)
        self.n_ops = self.L.bso_model_num_ops(self.h)
        self.n_tensors = self.L.bso_model_num_tensors(self.h)
        self.input = self.L.bso_model_input(self.h)
        self.output = self.L.bso_model_output(self.h)

    def shape(self, i):
        s = (C.c_int * 4)()
        rank = self.L.bso_model_tensor_shape(self.h, i, s)
        return tuple(s)[4 - rank:] if rank else ()

    def tensor(self, i) -> np.ndarray:
        n = self.L.bso_model_tensor_data(self.h, i, None, 0)
        a = np.empty(n, np.float32)
        self.L.bso_model_tensor_data(self.h, i, _f32(a), n)
        shp = self.shape(i)
        return a.reshape(shp) if int(np.prod(shp)) == n else a

    def op(self, i) -> dict:
        r = (C.c_int * 24)()
        self.L.bso_model_op(self.h, i, r)
        keys = ["code", "folded", "n_in", "in0", "in1", "in2", "in3", "out", "padding", "stride_w", "stride_h", "act",
                "dil_w", "dil_h", "depth_mult", "filter_w", "filter_h", "axis", "align_corners", "half_pixel"]
        return dict(zip(keys, list(r)))

    def invoke(self, x: np.ndarray) -> np.ndarray:
        x = np.ascontiguousarray(x, np.float32)
        out = np.empty(self.shape(self.output), np.float32)
        if self.L.bso_model_invoke(self.h, _f32(x), _f32(out)) != 0:
            raise RuntimeError("oracle invoke failed")
        return out

    def close(self):
        if self.owned and self.h:
            self.L.bso_model_free(self.h)
        self.h = None


class Ctx:
    """Mirror of bs_maskgen_new/process/delete on the CPU oracle."""

    def __init__(self, model_path, width, height, fast=False):
        self.L = lib(fast)
        self.h = self.L.bso_ctx_new(model_path.encode(), width, height)
        if not self.h:
            raise RuntimeError("oracle: cannot create context for %s" % model_path)
        g = (C.c_int * 15)()
        self.L.bso_ctx_geometry(self.h, g)
        g = list(g)
        (self.modeltype, self.inW, self.inH, self.inC, self.outW, self.outH, self.outC) = g[:7]
        self.roidim = tuple(g[7:11])
        self.in_roidim = tuple(g[11:15])
        self.width, self.height = width, height

    def process(self, frame: np.ndarray) -> np.ndarray:
        frame = np.ascontiguousarray(frame)
        mask = np.empty((self.height, self.width), np.uint8)
        if self.L.bso_ctx_process(self.h, _u8(frame), _u8(mask)) != 0:
            raise RuntimeError("oracle process failed")
        return mask

    def prep(self, frame):
        self.L.bso_ctx_prep(self.h, _u8(np.ascontiguousarray(frame)))
        return self.input()

    def infer(self):
        assert self.L.bso_ctx_infer(self.h) == 0
        return self.output()

    def set_output(self, logits):
        self.L.bso_ctx_set_output(self.h, _f32(np.ascontiguousarray(logits, np.float32)))

    def post(self):
        self.L.bso_ctx_post(self.h)
        return self.mask()

    def input(self):
        return np.ctypeslib.as_array(self.L.bso_ctx_input(self.h), (self.inH, self.inW, self.inC)).copy()

    def output(self):
        return np.ctypeslib.as_array(self.L.bso_ctx_output(self.h), (self.outH, self.outW, self.outC)).copy()

    def ofinal(self):
        return np.ctypeslib.as_array(self.L.bso_ctx_ofinal(self.h), (self.outH, self.outW)).copy()

    def set_ofinal(self, v):
        self.L.bso_ctx_set_ofinal(self.h, _u8(np.ascontiguousarray(v)))

    def mask(self):
        return np.ctypeslib.as_array(self.L.bso_ctx_mask(self.h), (self.height, self.width)).copy()

    def model(self):
        return Model(None, handle=self.L.bso_ctx_model(self.h))

    def close(self):
        if self.h:
            self.L.bso_ctx_delete(self.h)
        self.h = None


def baseline_run(model_path, frames: np.ndarray, bg: np.ndarray, iters: int, threads: int):
    """frames [S,H,W,3] u8, bg [H,W,3] u8 → (seconds, stage_seconds[4], out[S,H,W,3])"""
    L = lib(fast=True)
    frames = np.ascontiguousarray(frames)
    bg = np.ascontiguousarray(bg)
    S, H, W, _ = frames.shape
    out = np.empty_like(frames)
    st = (C.c_double * 4)()
    sec = L.bso_baseline_run(model_path.encode(), W, H, S, iters, threads, _u8(frames), _u8(bg), _u8(out), st)
    if sec < 0:
        raise RuntimeError("oracle baseline failed")
    return sec, list(st), out
