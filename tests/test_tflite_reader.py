"""The product's C++ .tflite loader + planner, exercised on the host only (bsx_model_describe needs no GPU)."""
import os
import re

import pytest

from conftest import MODEL_KEYS, ROOT, model_path, synthetic_model_path

# MMAC per frame of the four shipped graphs
MACS = {"lite": 14.0e6, "full": 33.5e6, "mlkit": 59.2e6, "deeplab": 725.8e6}


@pytest.fixture(scope="module")
def bs():
    from backscrub_amd import build
    build.build()
    import backscrub_amd
    return backscrub_amd


@pytest.mark.parametrize("key", list(MODEL_KEYS))
@pytest.mark.parametrize("real", [True, False])
def test_loader_and_planner_agree_with_python_reader(bs, key, real):
    from backscrub_amd import tflite_io
    path = model_path(key, prefer_real=real)
    m = tflite_io.load(path)
    txt = bs.model_describe(path)
    head = dict(kv.split("=") for kv in txt.splitlines()[0].split())
    assert int(head["ops"]) == len(m.ops)
    assert int(head["nodes"]) == sum(1 for o in m.ops if o.name != "DEQUANTIZE")
    # the planner may only REMOVE work (pw∘resize → resize∘pw runs the 1x1 conv at the low resolution)
    assert 0.85 * MACS[key] <= float(head["macs"]) <= 1.01 * MACS[key]
    steps = [l for l in txt.splitlines()[1:] if l[:4].strip().isdigit()]
    assert int(head["steps"]) == len(steps) < int(head["nodes"])       # fusion happened
    # every activation op was folded into its producer in the Meet/MLKit graphs
    if key != "deeplab":
        assert not any(re.search(r"\bact#", s) for s in steps)
        assert sum("scale=-1" not in s for s in steps) >= 5            # SE multiplies folded into the projection convs


def test_unknown_or_corrupt_files_fail_cleanly(bs, tmp_path):
    good = open(synthetic_model_path("lite"), "rb").read()
    with pytest.raises(bs.BsxError, match="unable to load model"):
        bs.model_describe(str(tmp_path / "missing.tflite"))
    for cut in (0, 3, 15, 64, 1000, len(good) // 2, len(good) - 1):
        p = tmp_path / ("cut%d.tflite" % cut)
        p.write_bytes(good[:cut])
        with pytest.raises(bs.BsxError):
            bs.model_describe(str(p))
    # flipped bytes in the header region must never crash the process
    import random
    rnd = random.Random(1)
    for k in range(40):
        b = bytearray(good)
        for _ in range(8):
            b[rnd.randrange(0, 4096)] = rnd.randrange(256)
        p = tmp_path / ("fuzz%d.tflite" % k)
        p.write_bytes(bytes(b))
        try:
            bs.model_describe(str(p))
        except bs.BsxError:
            pass


def test_unsupported_operator_is_reported(bs, tmp_path):
    from backscrub_amd import tflite_io as T
    from tools import make_synthetic_model as S
    m = S.build("lite")
    for op in m.ops:
        if op.name == "HARD_SWISH":
            op.code, op.name = 25, "SOFTMAX"   # a builtin the path does not implement
            break
    p = tmp_path / "segm_softmax.tflite"
    T.save(m, str(p))
    with pytest.raises(bs.BsxError, match="unsupported builtin operator code 25"):
        bs.model_describe(str(p))


@pytest.mark.parametrize("key", ["lite", "full", "mlkit", "deeplab"])
@pytest.mark.parametrize("real", [True, False])
def test_frame_program_lds_reservations_do_not_collide(key, real):
    """Host-only check of the per-frame program's lowering (plan.cpp verify_program_lds): every LDS reservation — tensors, DMA
    weight slots, band workspaces — with its lifetime; no two that are alive together may overlap, all inside the 160 KiB block."""
    import backscrub_amd
    path = model_path(key, prefer_real=real)
    if real and "synthetic" in os.path.basename(path):
        pytest.skip("reference model not staged on this box")
    line = [l for l in backscrub_amd.model_describe(path).splitlines() if l.startswith("program micro-ops=")][0]
    fields = dict(kv.split("=") for kv in line.split()[1:])
    if key == "deeplab" and int(fields["micro-ops"]) == 0:
        return        # DeepLab runs one launch per step (its ASPP pool branch is folded into a per-frame bias, a per-launch-only form): no program
    assert fields["lds_check"] == "ok", line
    assert int(fields["micro-ops"]) > 0 and int(fields["lds_blocks"]) >= int(fields["lds_tensors"])
    assert int(fields["lds_floats"]) <= 160 * 256


@pytest.mark.parametrize("real", [True, False])
def test_deeplab_plan_fuses_the_head_and_every_expand_depthwise_pair(real, monkeypatch, debug_switches):
    """Host-only check of the per-launch planner (plan.cpp): DeepLab's first three layers form the tiled head kernel, all 16 expand 1x1 → depthwise 3x3
    pairs fuse, every fused pair's LDS geometry (plan.hpp ir_geometry) fits — whole frame at 33x33, row bands of <= 78 KB (two workgroups per CU) above —
    and the switches that turn the fusions off produce the plain per-layer plan."""
    import re

    import backscrub_amd
    path = model_path("deeplab", prefer_real=real)
    if real and "synthetic" in os.path.basename(path):
        pytest.skip("reference model not staged on this box")
    for k in ("BSX_NO_IR_FUSE", "BSX_NO_HEAD0", "BSX_IR_GEOM", "BSX_H0_BH", "BSX_NO_CHAIN3"):
        monkeypatch.delenv(k, raising=False)
    text = backscrub_amd.model_describe(path)
    assert "fused with steps 1 and 2 (stem + depthwise + 1x1 in one tiled kernel)" in text
    pairs = re.findall(r"\^ fused with step (\d+) \(expand \+ depthwise in one kernel: (\d+) channels x (\d+) rows per workgroup, (\d+) band", text)
    assert len(pairs) == 16, text
    steps = {int(m.group(1)): m for m in re.finditer(r"^ *(\d+) (\w+) +\S+ +in (\d+)x(\d+)x(\d+) -> out (\d+)x(\d+)x(\d+) k\dx\d s(\d) d(\d)", text, re.M)}
    for dw_step, ch, bh, nb in pairs:
        m = steps[int(dw_step)]
        H, W, C, OH, S, d = int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)), int(m.group(9)), int(m.group(10))
        ch, bh, nb = int(ch), int(bh), int(nb)
        assert m.group(2) == "dwconv" and C % ch == 0 and ch in (16, 24, 32)
        assert nb * bh >= OH and (nb - 1) * bh < OH                       # the bands cover the output rows exactly once
        rows = min(H, S * (bh - 1) + 2 * d + 1)
        lds = rows * W * ch * 4
        assert lds <= (150 if nb == 1 else 78) * 1024, (dw_step, lds)
    # the ASPP head — 160 -> 256 (relu) -> 256 (relu, the pooled branch as its per-frame bias) -> 21 — is marked as one chained launch (pw_chain3_k), at the
    # middle convolution (the per-frame bias exists by then); the pool-branch GEMV sits between the first and the middle step
    m = re.search(r"\^ chained with steps (\d+) and (\d+)", text)
    assert m and text.count("chained with steps") == 1, text
    first, last = int(m.group(1)), int(m.group(2))
    mid = [i for i in steps if first < i < last and "-pool " in steps[i].group(0)]
    assert len(mid) == 1 and last == mid[0] + 1 and first == mid[0] - 2
    shape = lambda i: (int(steps[i].group(5)), int(steps[i].group(8)))
    assert [shape(i) for i in (first, mid[0], last)] == [(160, 256), (256, 256), (256, 21)]
    monkeypatch.setenv("BSX_NO_IR_FUSE", "1")
    monkeypatch.setenv("BSX_NO_HEAD0", "1")
    monkeypatch.setenv("BSX_NO_CHAIN3", "1")
    plain = backscrub_amd.model_describe(path)
    assert "fused with" not in plain and "chained with" not in plain and len(plain.splitlines()) < len(text.splitlines())


# ---- hostile / unsupported files: an error string, never a crash, an exception across the C ABI or a wrong network ----
def _describe_bytes(tmp_path, data, name="segm_hostile.tflite"):
    import ctypes

    import backscrub_amd
    p = tmp_path / name
    p.write_bytes(bytes(data))
    buf = ctypes.create_string_buffer(1 << 16)
    rc = backscrub_amd.lib().bsx_model_describe(str(p).encode(), buf, len(buf))
    return rc, buf.value.decode(errors="replace")


def test_loader_survives_mutated_files(tmp_path):
    """400 seeded mutations of a valid model (single bytes, 0xFFFFFFFF words = -1 indices / huge lengths, random words):
    bsx_model_describe must return 0 or BSX_EMODEL every time."""
    import numpy as np
    from tools import make_synthetic_model
    src = open(make_synthetic_model.ensure("lite"), "rb").read()
    rng = np.random.default_rng(5)
    seen = set()
    for _ in range(400):
        b = bytearray(src)
        for _ in range(int(rng.integers(1, 4))):
            region = int(rng.integers(0, 3))
            pos = int(rng.integers(0, 64)) if region == 0 else int(rng.integers(len(b) - 20000, len(b))) if region == 1 else int(rng.integers(0, len(b)))
            kind = int(rng.integers(0, 3))
            if kind == 0 or pos + 4 > len(b):
                b[pos] = int(rng.integers(0, 256))
            elif kind == 1:
                b[pos:pos + 4] = (0xFFFFFFFF).to_bytes(4, "little")
            else:
                b[pos:pos + 4] = int(rng.integers(0, 1 << 31)).to_bytes(4, "little")
        rc, _ = _describe_bytes(tmp_path, b)
        seen.add(rc)
        assert rc in (0, -2)
    assert seen == {0, -2}      # the mutations really did reach both outcomes


def test_unsupported_fused_activation_and_missing_operands_are_rejected(tmp_path):
    """A conv whose options carry TANH (4) / RELU_N1_TO_1 (2), a CONCATENATION with a fused activation, or an operator whose
    mandatory data operand is -1 must fail the load with a reason — not run as a different network / read tensors[-1]."""
    import copy

    from backscrub_amd import tflite_io as T
    from tools import make_synthetic_model
    base = make_synthetic_model.build("deeplab")      # DeepLab keeps its activations fused in the conv options

    def describe(model, name="deeplab_hostile.tflite"):
        path = str(tmp_path / name)
        T.save(model, path)
        return _describe_bytes(tmp_path, open(path, "rb").read(), name)

    rc, msg = describe(base)
    assert rc == 0, msg
    conv = next(i for i, o in enumerate(base.ops) if o.name == "CONV_2D" and o.opts.get("act", 0) in (1, 3))
    for bad in (2, 4, 5):
        m = copy.deepcopy(base)
        m.ops[conv].opts["act"] = bad
        rc, msg = describe(m)
        assert rc == -2 and "fused activation" in msg, msg
    cat = next(i for i, o in enumerate(base.ops) if o.name == "CONCATENATION")
    m = copy.deepcopy(base)
    m.ops[cat].opts["act"] = 1
    rc, msg = describe(m)
    assert rc == -2 and "CONCATENATION" in msg, msg
    for victim in ("CONV_2D", "DEPTHWISE_CONV_2D", "ADD", "RESIZE_BILINEAR", "AVERAGE_POOL_2D"):
        k = next(i for i, o in enumerate(base.ops) if o.name == victim)
        for operand in range(2 if victim != "AVERAGE_POOL_2D" else 1):
            m = copy.deepcopy(base)
            m.ops[k].inputs[operand] = -1
            rc, msg = describe(m)
            assert rc == -2 and "mandatory input" in msg, (victim, operand, msg)
