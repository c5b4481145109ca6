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


# ------------------------------------------------------------------------------------------------------------------------------
# Decision-margin audit of the arithmetic that cannot be pinned (TFLite / OpenCV are absent: SURVEY §8c "parity unpinned" rows)
# ------------------------------------------------------------------------------------------------------------------------------
def _scenes(W, H):
    """the two real webcam frames of tests/golden + synthetic scenes"""
    from backscrub_amd import synth
    from tools import make_photo_fixture
    out = [("photo0", f) for f in make_photo_fixture.load_frames()[:1]] if (W, H) == (640, 480) else []
    out += [("synthetic%d" % s, synth.frame(W, H, s, 0)) for s in (0, 3)]
    return out


def _decision_margin(modeltype_name, logits):
    """distance of every model-resolution pixel from its decision boundary, in the units the decision is taken in (lib/libbackscrub.cc:318-357)"""
    if modeltype_name == "meet":                     # e0/(e0+e1) < e1/(e0+e1)  ≡  l1 > l0
        return np.abs(logits[..., 1] - logits[..., 0])
    if modeltype_name == "mlkit":                    # p > 0.65
        return np.abs(logits[..., 0] - 0.65)
    person = logits[..., 15]                         # argmax == 15: gap between the person class and the best other class
    other = np.delete(logits, 15, axis=-1).max(-1)
    return np.abs(person - other)


AUDIT = {"lite": ("meet", (640, 480)), "full": ("meet", (1280, 720)), "mlkit": ("mlkit", (1280, 720)), "deeplab": ("deeplab", (640, 480))}


@pytest.mark.parametrize("key", list(AUDIT))
def test_decision_margin_audit(oracle, key):
    """How many pixels could the REAL reference (TFLite + XNNPACK: another f32 evaluation order of the same graph) decide differently from the oracle?
    For each real model on a webcam photo and synthetic scenes: (1) the network in float64 PyTorch from an independent parse of the file (tests/f64_graph.py) bounds
    the oracle's f32 logit error E; (2) pixels whose decision margin is below 10 E are the only ones another f32 evaluation can flip; (3) flipping ALL of them against
    the oracle gives the worst-case full-resolution mask after IIR + upscale + blur, and its IoU with the oracle's mask is the floor for "IoU vs the real reference".
    The numbers are recorded in DESIGN.md §2; the assertions keep them from silently degrading."""
    import os
    from conftest import MODEL_KEYS, ROOT
    import f64_graph
    path = os.path.join(ROOT, "models", MODEL_KEYS[key])
    if not os.path.exists(path):
        pytest.skip("real weights not staged (models/): the audit is about the shipped models")
    kind, (W, H) = AUDIT[key]
    report = []
    for name, frame in _scenes(W, H):
        ctx = oracle.Ctx(path, W, H)
        x = ctx.prep(frame)
        got = ctx.infer().astype(np.float64)
        val, m = f64_graph.run(path, x[None])
        want = val[m.outputs[0]][0].numpy()
        assert want.shape == got.shape
        err = float(np.abs(got - want).max())
        scale = max(1.0, float(np.abs(want).max()))
        assert err / scale < 1e-4, "%s/%s: oracle logits off by %g" % (key, name, err)          # whole-network check of the restatement, independent parse included
        margin = _decision_margin(kind, got)
        thr = 10.0 * err
        unsure = margin < thr
        # the decision of the float64 network must agree with the oracle wherever the margin is above the oracle's own error
        margin64 = _decision_margin(kind, want)
        if kind == "meet":
            dec, dec64 = got[..., 1] > got[..., 0], want[..., 1] > want[..., 0]
        elif kind == "mlkit":
            dec, dec64 = got[..., 0] > 0.65, want[..., 0] > 0.65
        else:
            dec, dec64 = got.argmax(-1) == 15, want.argmax(-1) == 15
        assert not np.any((dec != dec64) & (margin > 2 * err) & (margin64 > 2 * err))
        # worst case: every unsure pixel decided the other way, three frames so that the IIR has flushed, then upscale + blur
        ctx.close()
        a, b = oracle.Ctx(path, W, H), oracle.Ctx(path, W, H)
        flipped = got.astype(np.float32).copy()
        if kind == "meet":
            sw = flipped[unsure][:, ::-1].copy()
            eq = sw[:, 0] == sw[:, 1]
            sw[eq, 1] += 1.0                                    # ties decide "background": make them person instead
            flipped[unsure] = sw
        elif kind == "mlkit":
            flipped[unsure, 0] = np.where(got[unsure, 0] > 0.65, 0.0, 1.0)
        else:
            f = flipped[unsure]
            is_p = f.argmax(-1) == 15
            f[is_p, 15] = -1e4
            f[~is_p, 15] = 1e4
            flipped[unsure] = f
        for _ in range(3):
            a.prep(frame); a.set_output(got.astype(np.float32)); ma = a.post()
            b.prep(frame); b.set_output(flipped); mb = b.post()
        fa, fb = ma < 128, mb < 128
        union = np.logical_or(fa, fb).sum()
        iou = 1.0 if union == 0 else float(np.logical_and(fa, fb).sum() / union)
        a.close(); b.close()
        report.append({"scene": name, "oracle_logit_err_vs_f64": err, "threshold": thr, "model_pixels": int(margin.size), "pixels_below_threshold": int(unsure.sum()),
                       "person_fraction": float(dec.mean()), "full_res_pixels_changed": int((fa != fb).sum()), "worst_case_iou": iou})
    print("decision-margin audit %s: %s" % (key, report))
    for r in report:
        assert r["pixels_below_threshold"] <= 2e-3 * r["model_pixels"], r
        if r["person_fraction"] > 0.02:
            assert r["worst_case_iou"] >= 0.999, r
    out = os.environ.get("BSX_AUDIT_OUT")
    if out:
        import json
        with open(out, "a") as f:
            f.write(json.dumps({"model": key, "frame": [W, H], "scenes": report}) + "\n")
