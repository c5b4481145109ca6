"""Pins the oracle to the REFERENCE'S OWN OBJECT CODE wherever the reference's arithmetic is in its tree.

oracle/_ref/libbs_ref.so = /root/reference/lib/libbackscrub.cc + lib/transpose_conv_bias.cc + app/deepseg.cc:87-134,
compiled unmodified (oracle/Makefile `ref-lib`) against API shims whose OpenCV / TFLite-builtin operations are served by
the oracle's restatement.  So every comparison below isolates reference code: bs_maskgen_* glue and geometry, the decode
loops + IIR, Convolution2DTransposeBias Prepare/Eval, alpha_blend, the YUYV packing loop.  Rows pinned: SURVEY §8 a1, a2,
a10-a14, a17, a19, a20, f1 (packing).  NOT pinned by this (third-party, absent from the checkout): cv::resize, cvtColor,
bilateralFilter, blur, convertTo and the TFLite builtin kernels (rows a3-a9, a15, a16, a18)."""
import os
import shutil
import tempfile

import numpy as np
import pytest

from conftest import model_path
from oracle import oracle_py, ref_py

pytestmark = pytest.mark.skipif(not ref_py.available(), reason="oracle/_ref/libbs_ref.so absent and no reference checkout to build it from")

GEOMS = [("mlkit", 640, 480), ("lite", 640, 480), ("mlkit", 1280, 720), ("full", 1280, 720), ("lite", 1280, 720), ("full", 640, 480),
         ("lite", 322, 242), ("lite", 160, 96)]


def _frames(w, h, n, seed):
    from backscrub_amd import synth
    return [synth.frame(w, h, seed, t) for t in range(n)]


@pytest.mark.parametrize("key,w,h", GEOMS)
def test_whole_process_matches_reference_object_code(key, w, h):
    """bs_maskgen_new + 3 x bs_maskgen_process of the reference vs the oracle's Ctx: identical masks (geometry, state,
    decode+IIR, tconv inside the network, call order) on the SURVEY §8 frame x model pairs and two odd sizes."""
    path = model_path(key)
    ref = ref_py.RefMaskGen(path, w, h)
    assert ref.ok()
    oc = oracle_py.Ctx(path, w, h)
    for f in _frames(w, h, 3, seed=7):
        got, want = ref.process(f), oc.process(f)
        assert got is not None
        assert np.array_equal(got, want)
        q = oc.in_roidim
        assert np.array_equal(ref.last_ofinal(), oc.ofinal()[q[1]:q[1] + q[3], q[0]:q[0] + q[2]])
    ref.close(); oc.close()


def test_whole_process_deeplab_one_frame():
    path = model_path("deeplab")
    ref = ref_py.RefMaskGen(path, 640, 480)
    oc = oracle_py.Ctx(path, 640, 480)
    f = _frames(640, 480, 1, seed=3)[0]
    assert np.array_equal(ref.process(f), oc.process(f))
    ref.close(); oc.close()


def _adversarial_logits(modeltype, shape, rng):
    n = int(np.prod(shape[:2]))
    if modeltype == 3:        # Meet: softmax-2 through expf — ties, 1-ulp gaps, overflow (inf/inf = NaN → 255), NaN, ±inf
        l0 = rng.normal(0, 4, n).astype(np.float32)
        l1 = l0.copy()
        k = n // 8
        l1[:k] = np.nextafter(l0[:k], np.float32(np.inf)); l1[k:2 * k] = np.nextafter(l0[k:2 * k], np.float32(-np.inf))
        l1[2 * k:3 * k] = l0[2 * k:3 * k] + rng.normal(0, 1e-4, k).astype(np.float32)
        l0[3 * k:4 * k] = rng.uniform(80, 100, k); l1[3 * k:4 * k] = rng.uniform(80, 100, k)      # expf overflow
        l0[4 * k:5 * k] = rng.uniform(-110, -90, k); l1[4 * k:5 * k] = rng.uniform(-110, -90, k)  # both underflow → 0/0
        l1[5 * k:6 * k] = rng.normal(0, 4, k)
        sp = np.array([np.inf, -np.inf, np.nan, 0.0, -0.0, 88.72284, 88.72283], np.float32)
        l0[6 * k:7 * k] = rng.choice(sp, k); l1[6 * k:7 * k] = rng.choice(sp, k)
        l1[7 * k:] = rng.normal(0, 4, n - 7 * k)
        return np.stack([l0, l1], -1).reshape(shape)
    if modeltype == 2:        # MLKit: p > 0.65 as a DOUBLE compare — floats around (float)0.65 and 0.65 itself
        p = rng.uniform(0, 1, n).astype(np.float32)
        f65 = np.float32(0.65)
        near = np.array([f65, np.nextafter(f65, np.float32(1)), np.nextafter(f65, np.float32(0)), np.nan, np.inf, -np.inf, 0.6500001, 0.6499999], np.float32)
        p[: n // 4] = rng.choice(near, n // 4)
        return p.reshape(shape)
    x = rng.normal(0, 3, (n, shape[2])).astype(np.float32)   # DeepLab: argmax, first max wins, init -10000
    k = n // 6
    x[:k, 15] = x[:k].max(1)                                  # person ties with an EARLIER or LATER class
    x[k:2 * k] = -20000.0                                     # everything below the -10000 initial value → maxpos stays 0
    x[2 * k:3 * k, 15] = np.nan
    x[3 * k:4 * k, 15] = 1e9
    x[4 * k:5 * k, 14] = x[4 * k:5 * k, 15] = 50.0            # tie 14 vs 15: 14 wins
    return x.reshape(shape)


@pytest.mark.parametrize("key", ["lite", "mlkit", "deeplab"])
def test_decode_and_iir_match_reference_loops(key):
    """lib/libbackscrub.cc:317-357 through the reference's object code on adversarial logits, 5 frames deep (the IIR state
    walks 0x00 → 0xE0 → 0xFC → 0xFF and back), vs oracle decode_iir; and the mask the reference builds from that state
    (ROI placement, persistent 255 border) vs the oracle's post stage."""
    path = model_path(key)
    w, h = 640, 480
    ref = ref_py.RefMaskGen(path, w, h)
    oc = oracle_py.Ctx(path, w, h)
    rng = np.random.default_rng(11)
    frame = _frames(w, h, 1, seed=1)[0]
    state = np.zeros((oc.outH, oc.outW), np.uint8)
    q = oc.in_roidim
    for t in range(5):
        logits = _adversarial_logits(oc.modeltype, (oc.outH, oc.outW, oc.outC), rng)
        with np.errstate(all="ignore"):
            mask = ref.process(frame, forced_output=logits)
        assert mask is not None
        state = oracle_py.decode_iir(oc.modeltype, logits, state)
        assert np.array_equal(ref.last_ofinal(), state[q[1]:q[1] + q[3], q[0]:q[0] + q[2]]), "frame %d" % t
        oc.set_output(logits)
        assert np.array_equal(mask, oc.post()), "frame %d" % t
        assert np.array_equal(oc.ofinal(), state)
    ref.close(); oc.close()


@pytest.mark.parametrize("H,W,Ci,Co,k,s,padding", [(48, 80, 16, 2, 2, 2, 1), (16, 16, 16, 1, 2, 2, 1), (5, 7, 4, 3, 3, 2, 1), (5, 7, 4, 3, 3, 2, 2),
                                                  (6, 5, 8, 2, 4, 2, 1), (4, 4, 3, 5, 3, 1, 1), (3, 9, 2, 2, 2, 3, 2), (7, 3, 5, 1, 5, 3, 1)])
def test_transpose_conv_bias_matches_reference_op(H, W, Ci, Co, k, s, padding):
    """Convolution2DTransposeBias: the reference's registered Prepare (output shape, :118-186) + Eval (padding arithmetic and
    scatter loops, :188-256 → :37-114) vs the oracle restatement — shipped shapes (k = s = 2, SAME) and overlapping /
    padded / VALID geometries the shipped models never exercise.  Bit-exact: same loop order, same f32 accumulation."""
    rng = np.random.default_rng(H * 131 + W * 17 + k)
    x = rng.normal(0, 1, (1, H, W, Ci)).astype(np.float32)
    w = rng.normal(0, 0.5, (Co, k, k, Ci)).astype(np.float32)
    b = rng.normal(0, 1, Co).astype(np.float32)
    got = ref_py.tconv_bias(x, w, b, padding, (s, s))
    want = oracle_py.tconv_bias(x, w, b, padding, (s, s))
    assert got.shape == want.shape
    assert np.array_equal(got, want)


def test_alpha_blend_all_byte_triples_match_reference():
    """app/deepseg.cc:108-134 on all 2^24 (bg, frame, mask) byte triples: reference object code == oracle."""
    a, b, m = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    m2 = m.reshape(4096, 4096)
    # channel 0 enumerates every (bg=a, frame=b, m); the other channels carry other pairings of the same mask byte
    bg = np.stack([a.reshape(4096, 4096), b.reshape(4096, 4096), (a ^ b).reshape(4096, 4096)], -1)
    fr = np.stack([b.reshape(4096, 4096), a.reshape(4096, 4096), (255 - a).reshape(4096, 4096)], -1)
    assert np.array_equal(ref_py.alpha_blend(bg, fr, m2), oracle_py.alpha_blend(bg, fr, m2))


@pytest.mark.parametrize("w,h", [(640, 480), (2, 2), (322, 242), (6, 1)])
def test_yuyv_packing_matches_reference(w, h):
    """convert_rgb_to_yuyv (app/deepseg.cc:87-106): split + 4:2:2 packing (Y0 V Y1 U, truncating chroma mean) are the
    reference's; the RGB2YUV colour matrix underneath is the oracle's OpenCV restatement on both sides."""
    rng = np.random.default_rng(w * 7 + h)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    edge = np.array([[255, 255, 255], [0, 0, 0], [255, 0, 0], [0, 255, 0], [0, 0, 255], [254, 1, 128]], np.uint8)
    img[0, : min(w, 6)] = edge[: min(w, 6)]
    assert np.array_equal(ref_py.convert_rgb_to_yuyv(img), oracle_py.bgr_to_yuyv(img))


def test_reference_interface_behaviour():
    """Error / callback / aliasing behaviour of the reference itself, which tests/test_shim.py and test_cabi.py assert for
    the product: NULL on a missing file and on an unknown model name, false on a NULL context, callbacks in the order
    prep → infer → mask once per call, mask header aliasing one lib-owned buffer."""
    msgs = []
    assert not ref_py.RefMaskGen("/nonexistent/segm_x.tflite", 640, 480, ondebug=lambda c, m: msgs.append(m)).ok()
    assert msgs and b"unable to load model" in msgs[0]
    d = tempfile.mkdtemp()
    odd = os.path.join(d, "mystery.tflite")
    shutil.copy(model_path("lite"), odd)
    msgs.clear()
    assert not ref_py.RefMaskGen(odd, 640, 480, ondebug=lambda c, m: msgs.append(m)).ok()
    assert any(b"unknown model type" in m for m in msgs)
    shutil.rmtree(d)
    order = []
    ref = ref_py.RefMaskGen(model_path("lite"), 640, 480, onprep=lambda c: order.append("prep"), oninfer=lambda c: order.append("infer"),
                            onmask=lambda c: order.append("mask"))
    f = _frames(640, 480, 1, seed=2)[0]
    ref.process(f); p1 = ref.last_mask_ptr
    ref.process(f); p2 = ref.last_mask_ptr
    assert order == ["prep", "infer", "mask"] * 2
    assert p1 == p2 and p1
    ref.close()
    out = np.empty((480, 640), np.uint8)
    assert ref_py.lib().ref_maskgen_process(None, f.ctypes.data_as(ref_py._u8p), 640, 480, out.ctypes.data_as(ref_py._u8p), None) == 0
    assert b"2.8.0" in ref_py.lib().ref_tensorflow_version()
