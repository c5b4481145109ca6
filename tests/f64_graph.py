"""Float64 evaluation of a .tflite graph with PyTorch CPU — TEST INFRASTRUCTURE, independent of the oracle twice over: the file is parsed by the Python reader
(backscrub_amd/tflite_io.py, not the oracle's C++ parser) and every operator is a PyTorch float64 op (not the oracle's loops).  Used by the decision-margin audit
(tests/test_oracle_nn.py) as the "exact" network: |oracle_f32 - f64| measures the rounding error of the oracle's f32 restatement, and — TFLite's real runtime path
(XNNPACK) being another f32 evaluation of the same graph with another summation order — is the yardstick for how far the real reference's logits can be from the
oracle's.  Semantics per SURVEY.md §8(c) (TFLite reference kernels): SAME padding with the extra pixel at the bottom / right, half-pixel / align-corners bilinear,
Convolution2DTransposeBias 2x2 stride 2 = conv_transpose2d."""
import numpy as np
import torch
import torch.nn.functional as F

from backscrub_amd import tflite_io as T


def _same_pad(inp, k, s, d):
    out = -(-inp // s)
    total = max(0, (out - 1) * s + (k - 1) * d + 1 - inp)
    return total // 2, total - total // 2


def _act(y, act):
    return {0: y, 1: F.relu(y), 3: torch.clamp(y, 0, 6)}[act]


def _nchw(a):
    return a.permute(0, 3, 1, 2)


def _nhwc(a):
    return a.permute(0, 2, 3, 1)


def _resize(x, oh, ow, align, half):
    """TFLite RESIZE_BILINEAR (reference kernel) in float64: in = (o + 0.5) * scale - 0.5 (half_pixel) or o * (in - 1) / (out - 1) (align_corners);
    lo = max(floor, 0), hi = min(ceil, in - 1), fraction from the un-clamped floor."""
    n, h, w, c = x.shape

    def axis(o, i):
        idx = torch.arange(o, dtype=torch.float64)
        if align and o > 1:
            src = idx * ((i - 1) / (o - 1))
        elif half:
            src = (idx + 0.5) * (i / o) - 0.5
        else:
            src = idx * (i / o)
        fl = torch.floor(src)
        lo = torch.clamp(fl, min=0).long()
        hi = torch.clamp(torch.ceil(src), max=i - 1).long()
        hi = torch.maximum(hi, torch.zeros_like(hi))
        return lo, hi, src - fl
    y0, y1, fy = axis(oh, h)
    x0, x1, fx = axis(ow, w)
    fy = fy.view(1, oh, 1, 1)
    fx = fx.view(1, 1, ow, 1)
    top = x[:, y0][:, :, x0] * (1 - fx) + x[:, y0][:, :, x1] * fx
    bot = x[:, y1][:, :, x0] * (1 - fx) + x[:, y1][:, :, x1] * fx
    return top * (1 - fy) + bot * fy


def run(path, x_nhwc):
    """x_nhwc: [1,H,W,3] float (the f32 network input, exactly as the oracle's prep produced it) → dict tensor index → float64 torch tensor (NHWC) for every tensor."""
    m = T.load(path)
    val = {}
    for i, t in enumerate(m.tensors):
        if t.data is not None:
            val[i] = torch.from_numpy(np.asarray(t.data).astype(np.float64)) if t.type != T.TENSOR_I32 else torch.from_numpy(np.asarray(t.data))
    val[m.inputs[0]] = torch.from_numpy(np.asarray(x_nhwc, dtype=np.float64))
    for op in m.ops:
        i, o = op.inputs, op.outputs[0]
        name = op.name
        if name == "DEQUANTIZE":
            val[o] = val[i[0]]                                   # f16 constant → exact in f64
        elif name in ("CONV_2D", "DEPTHWISE_CONV_2D"):
            x = _nchw(val[i[0]])
            w = val[i[1]]
            b = val[i[2]] if len(i) > 2 and i[2] >= 0 else None
            if name == "CONV_2D":
                wt, groups = w.permute(0, 3, 1, 2), 1
            else:
                wt, groups = w.permute(3, 0, 1, 2), x.shape[1]
            kh, kw = wt.shape[2], wt.shape[3]
            sh, sw, dh, dw = op.opts["stride_h"], op.opts["stride_w"], op.opts["dil_h"], op.opts["dil_w"]
            if op.opts["padding"] == 0:
                pt, pb = _same_pad(x.shape[2], kh, sh, dh)
                pl, pr = _same_pad(x.shape[3], kw, sw, dw)
                x = F.pad(x, (pl, pr, pt, pb))
            val[o] = _nhwc(_act(F.conv2d(x, wt.contiguous(), b, stride=(sh, sw), dilation=(dh, dw), groups=groups), op.opts["act"]))
        elif name == "FULLY_CONNECTED":
            x = val[i[0]]
            y = F.linear(x.reshape(-1, x.shape[-1]), val[i[1]], val[i[2]] if len(i) > 2 and i[2] >= 0 else None)
            val[o] = _act(y.reshape(tuple(x.shape[:-1]) + (y.shape[-1],)) if op.opts.get("keep_num_dims") else y, op.opts["act"])
        elif name == "AVERAGE_POOL_2D":
            x = val[i[0]]
            assert op.opts["filter_h"] == x.shape[1] and op.opts["filter_w"] == x.shape[2]
            val[o] = x.mean((1, 2), keepdim=True)
        elif name in ("ADD", "MUL"):
            a, b = val[i[0]], val[i[1]]
            val[o] = _act(a + b if name == "ADD" else a * b, op.opts["act"])
        elif name == "RELU":
            val[o] = F.relu(val[i[0]])
        elif name == "RELU6":
            val[o] = torch.clamp(val[i[0]], 0, 6)
        elif name == "HARD_SWISH":
            v = val[i[0]]
            val[o] = v * torch.clamp(v + 3, 0, 6) / 6
        elif name == "LOGISTIC":
            val[o] = torch.sigmoid(val[i[0]])
        elif name == "CONCATENATION":
            val[o] = torch.cat([val[k] for k in i], dim=op.opts["axis"] if op.opts["axis"] >= 0 else op.opts["axis"] + 4)
        elif name == "RESIZE_BILINEAR":
            oh, ow = [int(v) for v in val[i[1]].reshape(-1)]
            val[o] = _resize(val[i[0]], oh, ow, bool(op.opts["align_corners"]), bool(op.opts["half_pixel_centers"]))
        elif op.code == 32:                                       # Convolution2DTransposeBias: 2x2 stride 2, SAME → no overlap
            x = _nchw(val[i[0]])
            w = val[i[1]].permute(3, 0, 1, 2)                     # [O,kh,kw,I] → [I,O,kh,kw]
            val[o] = _nhwc(F.conv_transpose2d(x, w.contiguous(), val[i[2]], stride=2))
        else:
            raise AssertionError("operator %s has no float64 mirror" % name)
    return val, m
