"""Emit architecture-faithful `.tflite` models with seeded random weights.

The reference's model files (`/root/reference/models/*.tflite`) are data that a user of
the drop-in supplies at run time; they do not exist on the GPU box.  So that the GPU parity
tests, `smoke()` and `bench.py` always have a model to run, this tool writes the four
supported architectures from scratch — same operators, shapes,
options, f16-weights-behind-DEQUANTIZE storage — with deterministic random weights:

    lite    segm_lite_v681      96x160x3  -> 96x160x2   (Google Meet, MobileNetV3-small-ish)
    full    segm_full_v679      144x256x3 -> 144x256x2
    mlkit   selfiesegmentation_mlkit-256x256 (f16)  256x256x3 -> 256x256x1
    deeplab deeplabv3_257_mv_gpu  257x257x3 -> 257x257x21

File names keep the substrings the reference sniffs for the model type
(`lib/libbackscrub.cc:116-130`: "segm_", "selfie", "deeplab").
`tests/test_synthetic_models.py` checks op list + shapes against the real files when they
are available.
"""
from __future__ import annotations

import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from backscrub_amd import tflite_io as T  # noqa: E402

OUT_DIR = os.path.join(ROOT, "tests", "golden", "models")
FILES = {
    "lite": "synthetic_segm_lite_96x160.tflite",
    "full": "synthetic_segm_full_144x256.tflite",
    "mlkit": "synthetic_selfie_mlkit_256x256.f16.tflite",
    "deeplab": "synthetic_deeplabv3_257.tflite",
}
SAME, VALID = 0, 1
NONE, RELU, RELU6 = 0, 1, 3


class G:
    """tiny graph builder on top of tflite_io.Model"""

    def __init__(self, seed, f16_weights):
        self.rng = np.random.default_rng(seed)
        self.f16 = f16_weights
        self.t = []
        self.ops = []

    def tensor(self, shape, name, data=None, ttype=T.TENSOR_F32):
        self.t.append(T.Tensor(list(shape), ttype, 0, name, data))
        return len(self.t) - 1

    def shape(self, i):
        return self.t[i].shape

    def weight(self, arr, name):
        """constant weight: f16 constant + DEQUANTIZE (Google models) or plain f32 constant"""
        arr = np.asarray(arr, np.float32)
        if not self.f16:
            return self.tensor(arr.shape, name, arr)
        h = self.tensor(arr.shape, name + "_f16", arr.astype(np.float16), T.TENSOR_F16)
        o = self.tensor(arr.shape, name)
        self.ops.append(T.Op(T.OPCODES["DEQUANTIZE"], "DEQUANTIZE", [h], [o]))
        return o

    def _rand(self, shape, fan_in, gain):
        return self.rng.standard_normal(shape).astype(np.float32) * np.float32(gain / np.sqrt(fan_in))

    def op(self, name, ins, out_shape, opts=None, custom=b""):
        o = self.tensor(out_shape, "%s_%d" % (name.lower(), len(self.ops)))
        code = T.OPCODES.get(name, 32)
        self.ops.append(T.Op(code, name, list(ins), [o], opts or {}, custom))
        return o

    @staticmethod
    def _osz(i, k, s, d, pad):
        eff = (k - 1) * d + 1
        return -(-i // s) if pad == SAME else (i + s - eff) // s

    def conv(self, x, cout, k=1, stride=1, pad=SAME, act=NONE, gain=1.4, dil=1, bias_add=None):
        _, h, w, c = self.shape(x)
        wt = self.weight(self._rand((cout, k, k, c), k * k * c, gain), "w")
        bias = self.rng.uniform(-0.1, 0.1, cout)
        for ch, v in (bias_add or {}).items():
            bias[ch] += v
        b = self.weight(bias, "b")
        oh, ow = self._osz(h, k, stride, dil, pad), self._osz(w, k, stride, dil, pad)
        return self.op("CONV_2D", [x, wt, b], [1, oh, ow, cout],
                       dict(padding=pad, stride_w=stride, stride_h=stride, act=act, dil_w=dil, dil_h=dil))

    def dw(self, x, k=3, stride=1, act=NONE, gain=1.4, dil=1):
        _, h, w, c = self.shape(x)
        wt = self.weight(self._rand((1, k, k, c), k * k, gain), "dw")
        b = self.weight(self.rng.uniform(-0.1, 0.1, c), "b")
        oh, ow = self._osz(h, k, stride, dil, SAME), self._osz(w, k, stride, dil, SAME)
        return self.op("DEPTHWISE_CONV_2D", [x, wt, b], [1, oh, ow, c],
                       dict(padding=SAME, stride_w=stride, stride_h=stride, depth_mult=1, act=act, dil_w=dil, dil_h=dil))

    def fc(self, x, cout, gain=1.0):
        c = self.shape(x)[-1]
        wt = self.weight(self._rand((cout, c), c, gain), "fcw")
        b = self.weight(self.rng.uniform(-0.1, 0.1, cout), "b")
        return self.op("FULLY_CONNECTED", [x, wt, b], self.shape(x)[:-1] + [cout], dict(act=NONE, keep_num_dims=1))

    def unary(self, name, x):
        return self.op(name, [x], self.shape(x))

    def gap(self, x):
        _, h, w, c = self.shape(x)
        return self.op("AVERAGE_POOL_2D", [x], [1, 1, 1, c], dict(padding=VALID, stride_w=w, stride_h=h, filter_w=w, filter_h=h, act=NONE))

    def binary(self, name, a, b):
        sa, sb = self.shape(a), self.shape(b)
        return self.op(name, [a, b], [max(p, q) for p, q in zip(sa, sb)], dict(act=NONE))

    def concat(self, xs, axis=-1):
        s = list(self.shape(xs[0]))
        s[-1] = sum(self.shape(x)[-1] for x in xs)
        return self.op("CONCATENATION", xs, s, dict(axis=axis))

    def resize(self, x, oh, ow, align=0, half=1):
        sz = self.tensor([2], "size", np.array([oh, ow], np.int32), T.TENSOR_I32)
        return self.op("RESIZE_BILINEAR", [x, sz], [1, oh, ow, self.shape(x)[3]], dict(align_corners=align, half_pixel_centers=half))

    def tconv(self, x, cout, gain=2.0):
        _, h, w, c = self.shape(x)
        wt = self.weight(self._rand((cout, 2, 2, c), c, gain), "tw")
        b = self.weight(self.rng.uniform(-0.2, 0.2, cout), "tb")
        # TfLiteTransposeConvParams{padding=SAME(1), stride_w=2, stride_h=2}
        return self.op("Convolution2DTransposeBias", [x, wt, b], [1, 2 * h, 2 * w, cout], custom=struct.pack("<iii", 1, 2, 2))

    def model(self, inp, out, desc):
        return T.Model(self.t, self.ops, [inp], [out], desc)


def _meet_family(H, W, mlkit, seed):
    """B.1 (Meet lite/full) and B.2 (MLKit) share the encoder/decoder skeleton."""
    g = G(seed, f16_weights=True)
    R = "RELU" if mlkit else "RELU6"
    x = g.tensor([1, H, W, 3], "input")
    A = g.unary("HARD_SWISH", g.conv(x, 16, 3, 2))

    def se(t, squeeze):
        c = g.shape(t)[3]
        p = g.gap(t)
        if mlkit:   # 1x1 CONV (VALID) with channel squeeze
            s = g.unary("LOGISTIC", g.conv(g.unary("RELU", g.conv(p, c // squeeze, 1, 1, VALID)), c, 1, 1, VALID))
        else:       # FULLY_CONNECTED C->C
            s = g.unary("LOGISTIC", g.fc(g.unary("RELU", g.fc(p, c)), c))
        return g.binary("MUL", t, s)

    def ir(t, cexp, cout, k, stride, act, use_se, squeeze=4, residual=False):
        e = g.unary(act, g.conv(t, cexp))
        d = g.unary(act, g.dw(e, k, stride))
        if use_se:
            d = se(d, squeeze)
        o = g.conv(d, cout, gain=1.0)
        return g.binary("ADD", o, t) if residual else o

    B = ir(A, 16, 16, 3, 2, R, True, squeeze=2)
    t36 = ir(B, 72, 24, 3, 2, R, False)
    Cc = ir(t36, 88, 24, 3, 1, R, False, residual=True)
    t = ir(Cc, 96, 32, 5, 2, "HARD_SWISH", True)
    t = ir(t, 128, 32, 5, 1, "HARD_SWISH", True, residual=True)
    t = ir(t, 128, 32, 5, 1, "HARD_SWISH", True, residual=True)
    t = ir(t, 96, 32, 5, 1, "HARD_SWISH", True, residual=True)
    if mlkit:
        t = ir(t, 96, 32, 5, 1, "HARD_SWISH", True, residual=True)
        p = g.gap(t)
        c1 = g.conv(t, 128, 1, 1, VALID)
        gt = g.conv(p, 128, 1, 1, VALID)
    else:
        t = ir(t, 72, 24, 5, 1, "HARD_SWISH", True)
        c1 = g.conv(t, 128)
        p = g.gap(t)
        gt = g.conv(p, 128, 1, 1, VALID)
    head = g.binary("MUL", g.unary(R, c1), g.unary("LOGISTIC", gt))

    def dec(t, skip, cout):
        _, h, w, _ = g.shape(skip)
        u = g.conv(g.resize(t, h, w), cout, 1, 1, VALID)
        s = g.binary("ADD", skip, u) if mlkit else g.concat([skip, u])
        gate = g.unary("LOGISTIC", g.conv(g.unary("RELU", g.conv(g.gap(s), cout, 1, 1, VALID)), cout, 1, 1, VALID))
        y = g.binary("ADD", g.binary("MUL", skip, gate), u)
        z = g.conv(y, cout, 1, 1, VALID, gain=1.0)
        if mlkit:
            z = g.unary("RELU", z)
        return g.binary("ADD", z, g.unary(R, g.dw(z, 3, 1)))

    t = dec(head, Cc, 24)
    t = dec(t, B, 16)
    t = dec(t, A, 16)
    if mlkit:
        out = g.unary("LOGISTIC", g.tconv(t, 1))
    else:
        out = g.tconv(t, 2)
    return g.model(x, out, "synthetic %s %dx%d (seed %d)" % ("mlkit-selfie" if mlkit else "meet", H, W, seed))


def _deeplab(seed):
    g = G(seed, f16_weights=False)
    x = g.tensor([1, 257, 257, 3], "input")
    t = g.conv(x, 16, 3, 2, SAME, RELU6)
    t = g.conv(g.dw(t, 3, 1, RELU6), 8, gain=1.0)

    def ir(t, cexp, cout, stride=1, dil=1, residual=False):
        o = g.conv(g.dw(g.conv(t, cexp, act=RELU6), 3, stride, RELU6, dil=dil), cout, gain=1.0)
        return g.binary("ADD", o, t) if residual else o

    t = ir(t, 48, 12, 2)
    t = ir(t, 72, 12, residual=True)
    t = ir(t, 72, 16, 2)
    t = ir(t, 96, 16, residual=True)
    t = ir(t, 96, 16, residual=True)
    t = ir(t, 96, 32)
    for _ in range(3):
        t = ir(t, 192, 32, dil=2, residual=True)
    t = ir(t, 192, 48, dil=2)
    t = ir(t, 288, 48, dil=2, residual=True)
    t = ir(t, 288, 48, dil=2, residual=True)
    t = ir(t, 288, 80, dil=2)
    t = ir(t, 480, 80, dil=4, residual=True)
    t = ir(t, 480, 80, dil=4, residual=True)
    t = ir(t, 480, 160, dil=4)
    pool = g.resize(g.conv(g.gap(t), 256, act=RELU), 33, 33, align=1, half=0)
    br = g.conv(t, 256, act=RELU)
    t = g.conv(g.concat([pool, br], axis=3), 256, act=RELU)
    t = g.conv(t, 21, gain=3.0, bias_add={15: 6.0})  # let "person" win somewhere
    t = g.resize(t, 33, 33, align=1, half=0)
    out = g.resize(t, 257, 257, align=1, half=0)
    return g.model(x, out, "synthetic deeplabv3-mnv2 257 (seed %d)" % seed)


def build(key: str, seed: int = 1234) -> T.Model:
    if key == "lite":
        return _meet_family(96, 160, False, seed)
    if key == "full":
        return _meet_family(144, 256, False, seed + 1)
    if key == "mlkit":
        return _meet_family(256, 256, True, seed + 2)
    if key == "deeplab":
        return _deeplab(seed + 3)
    raise KeyError(key)


def ensure(key: str) -> str:
    """path of the synthetic model, generating it on first use"""
    path = os.path.join(OUT_DIR, FILES[key])
    if not os.path.exists(path):
        os.makedirs(OUT_DIR, exist_ok=True)
        T.save(build(key), path)
    return path


if __name__ == "__main__":
    for k in (sys.argv[1:] or list(FILES)):
        p = os.path.join(OUT_DIR, FILES[k])
        os.makedirs(OUT_DIR, exist_ok=True)
        T.save(build(k), p)
        print(k, p, os.path.getsize(p))
