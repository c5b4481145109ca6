"""Cross-check of the CPU oracle's NN kernels against CPU PyTorch, op by op.

The oracle restates TFLite's reference kernels; PyTorch is an independent implementation
of the same operators, so agreement (rtol 1e-4 — summation order differs) guards against
restatement bugs (SAME-padding asymmetry at stride 2, dilation, half-pixel bilinear, ...).
Every non-constant op of every available model is checked on the activations the oracle
itself produced for a synthetic frame.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import MODEL_KEYS, model_path

ADD, AVGPOOL, CONCAT, CONV, DW, DEQ, FC, LOGISTIC, MUL, RELU, RELU6, RESIZE, CUSTOM, HSWISH = 0, 1, 2, 3, 4, 6, 9, 14, 18, 19, 21, 23, 32, 117


def _same_pad(inp, k, s, d):
    out = -(-inp // s)
    total = max(0, (out - 1) * s + (k - 1) * d + 1 - inp)
    return total // 2, total - total // 2


def _nchw(a):
    return torch.from_numpy(np.ascontiguousarray(a)).permute(0, 3, 1, 2).double()


def _act(y, act):
    return {0: y, 1: F.relu(y), 3: torch.clamp(y, 0, 6)}[act]


def _torch_op(m, op):
    t = lambda k: m.tensor(op[k])
    c = op["code"]
    if c in (CONV, DW):
        x = _nchw(t("in0"))
        w = t("in1")
        b = torch.from_numpy(t("in2")).double() if op["in2"] >= 0 else None
        if c == CONV:
            wt = torch.from_numpy(w).permute(0, 3, 1, 2).double()
            groups = 1
        else:
            wt = torch.from_numpy(w).permute(3, 0, 1, 2).double()
            groups = x.shape[1]
        kh, kw = wt.shape[2], wt.shape[3]
        if op["padding"] == 0:
            pt, pb = _same_pad(x.shape[2], kh, op["stride_h"], op["dil_h"])
            pl, pr = _same_pad(x.shape[3], kw, op["stride_w"], op["dil_w"])
            x = F.pad(x, (pl, pr, pt, pb))
        y = F.conv2d(x, wt, b, stride=(op["stride_h"], op["stride_w"]), dilation=(op["dil_h"], op["dil_w"]), groups=groups)
        return _act(y, op["act"]).permute(0, 2, 3, 1)
    if c == FC:
        x = torch.from_numpy(t("in0")).double()
        y = F.linear(x, torch.from_numpy(t("in1")).double(), torch.from_numpy(t("in2")).double())
        return _act(y, op["act"])
    if c == AVGPOOL:
        x = _nchw(t("in0"))
        assert op["filter_h"] == x.shape[2] and op["filter_w"] == x.shape[3]
        return x.mean((2, 3), keepdim=True).permute(0, 2, 3, 1)
    if c in (ADD, MUL):
        a, b = torch.from_numpy(t("in0")).double(), torch.from_numpy(t("in1")).double()
        return _act(a + b if c == ADD else a * b, op["act"])
    if c == RELU:
        return F.relu(torch.from_numpy(t("in0")).double())
    if c == RELU6:
        return torch.clamp(torch.from_numpy(t("in0")).double(), 0, 6)
    if c == HSWISH:
        return F.hardswish(torch.from_numpy(t("in0")).double())
    if c == LOGISTIC:
        return torch.sigmoid(torch.from_numpy(t("in0")).double())
    if c == CONCAT:
        return torch.cat([torch.from_numpy(m.tensor(op[k])).double() for k in ("in0", "in1", "in2", "in3")[:op["n_in"]]], -1)
    if c == RESIZE:
        x = _nchw(t("in0"))
        oh, ow = m.shape(op["out"])[1:3]
        y = F.interpolate(x, size=(oh, ow), mode="bilinear", align_corners=bool(op["align_corners"]))
        if not op["align_corners"]:
            assert op["half_pixel"] == 1  # torch's align_corners=False is the half-pixel convention
        return y.permute(0, 2, 3, 1)
    if c == CUSTOM:
        # Convolution2DTransposeBias, k=s=2, pad 0  ==  conv_transpose2d
        x = _nchw(t("in0"))
        w = torch.from_numpy(t("in1")).permute(3, 0, 1, 2).double()  # [O,kh,kw,I] -> [I,O,kh,kw]
        y = F.conv_transpose2d(x, w, torch.from_numpy(t("in2")).double(), stride=2)
        return y.permute(0, 2, 3, 1)
    raise AssertionError("op code %d has no torch mirror" % c)


@pytest.mark.parametrize("key", list(MODEL_KEYS))
def test_oracle_ops_match_torch(oracle, key):
    from backscrub_amd import synth
    path = model_path(key)
    res = (640, 480)
    ctx = oracle.Ctx(path, *res)
    ctx.prep(synth.frame(*res, stream=1))
    ctx.infer()
    m = ctx.model()
    checked = 0
    for i in range(m.n_ops):
        op = m.op(i)
        if op["folded"] or op["code"] == DEQ:
            continue
        want = _torch_op(m, op).numpy()
        got = m.tensor(op["out"])
        assert got.shape == tuple(want.shape), (i, op, got.shape, want.shape)
        scale = max(1.0, float(np.abs(want).max()))
        err = float(np.abs(got - want).max()) / scale
        assert err < 2e-5, "op #%d code %d: rel err %g" % (i, op["code"], err)
        checked += 1
    assert checked > 50
    ctx.close()
