"""GPU parity tests: the HIP path (through the C ABI of libbsx.so) against the CPU oracle.

Bars (BASELINE.json north_star): integer/byte stages bit-exact; network logits within
float tolerance; end to end mask IoU >= 0.999 and composited frame max-abs <= 1 LSB.
"""
import os

import numpy as np
import pytest

from conftest import MODEL_KEYS, model_path, synthetic_model_path

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

VGA, HD = (640, 480), (1280, 720)


@pytest.fixture(scope="module")
def bs():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a visible MI355X (torch.cuda.is_available() is False)")
    import backscrub_amd
    backscrub_amd.lib()  # raises if libbsx.so is missing: no silent fallback
    return backscrub_amd


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _iou_fg(a, b, need_person=False):
    """IoU of the person region (mask < 128) — 1.0 when both are empty, unless the caller says the oracle mask (b) must
    contain a person (end-to-end tests on the photo fixture): an empty expectation then FAILS instead of passing vacuously."""
    fa, fb = a < 128, b < 128
    if need_person:
        assert fb.mean() > 0.05, "oracle mask has no person region (%.4f): the test would be vacuous" % fb.mean()
    union = np.logical_or(fa, fb).sum()
    return 1.0 if union == 0 else np.logical_and(fa, fb).sum() / union


# --------------------------------------------------------------------------------------------
# alpha blend (deepseg.cc:108-134): bit exact, including every (a, b, m) byte combination
# --------------------------------------------------------------------------------------------
def test_blend_exhaustive_identity(bs, oracle):
    m, a, b = np.meshgrid(np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), np.arange(256, dtype=np.uint8), indexing="ij")
    W, H = 4096, 4096  # 2^24 pixels = all (m,a,b) triples, one per pixel
    mask = m.reshape(H, W)
    bg = np.repeat(a.reshape(H, W, 1), 3, 2)
    fr = np.repeat(b.reshape(H, W, 1), 3, 2)
    mg = bs.MaskGen(synthetic_model_path("lite"), W, H, n_streams=1)
    out = mg.composite(_dev(bg), _dev(fr[None]), _dev(mask[None]))[0].cpu().numpy()
    want = ((bg.astype(np.int32) * mask[..., None] + fr.astype(np.int32) * (255 - mask[..., None].astype(np.int32))) // 255).astype(np.uint8)
    assert np.array_equal(out, want)
    assert np.array_equal(want[:64], oracle.alpha_blend(bg[:64], fr[:64], mask[:64]))
    mg.close()


@pytest.mark.parametrize("res,n,shared_bg", [(VGA, 5, True), (HD, 2, False), ((322, 243), 3, True)])
def test_blend_matches_oracle(bs, oracle, res, n, shared_bg):
    from backscrub_amd import synth
    W, H = res
    fr = synth.random_u8((n, H, W, 3), 1)
    bg = synth.random_u8((H, W, 3) if shared_bg else (n, H, W, 3), 2)
    mask = synth.random_u8((n, H, W), 3)
    mask[0, :8] = 255
    mask[0, 8:16] = 0
    mg = bs.MaskGen(synthetic_model_path("lite"), W, H, n_streams=n)
    out = mg.composite(_dev(bg), _dev(fr), _dev(mask)).cpu().numpy()
    for i in range(n):
        assert np.array_equal(out[i], oracle.alpha_blend(bg if shared_bg else bg[i], fr[i], mask[i])), "frame %d" % i
    mg.close()


# --------------------------------------------------------------------------------------------
# stage-by-stage parity on identical inputs
# --------------------------------------------------------------------------------------------
CASES = [("lite", VGA), ("lite", HD), ("full", HD), ("full", VGA), ("mlkit", VGA), ("mlkit", HD), ("deeplab", VGA)]


@pytest.mark.parametrize("key,res", CASES)
@pytest.mark.parametrize("real", [True, False])
def test_stages_match_oracle(bs, oracle, key, res, real):
    from backscrub_amd import synth
    path = model_path(key, prefer_real=real)
    if real and "synthetic" in os.path.basename(path):
        pytest.skip("reference model not staged on this box")
    W, H = res
    n = 3
    mg = bs.MaskGen(path, W, H, n_streams=n)
    oc = [oracle.Ctx(path, W, H) for _ in range(n)]
    info = mg.info
    assert tuple(info["roi"]) == oc[0].roidim and tuple(info["in_roi"]) == oc[0].in_roidim
    frames = np.stack([synth.frame(W, H, s) for s in range(n)])
    frames[2] = synth.random_u8((H, W, 3), 7)  # one pure-noise stream stresses the integer paths
    d_frames = _dev(frames)

    # (0) prep: resize + BGR2RGB + bilateral + normalise — bit exact
    mg.run_stage(0, d_frames)
    got_in = mg.input_tensor().cpu().numpy()
    for i in range(n):
        want = oc[i].prep(frames[i])
        assert np.array_equal(got_in[i], want), "prep mismatch stream %d: %d px differ, max %g" % (
            i, (got_in[i] != want).sum(), np.abs(got_in[i] - want).max())

    # (1) network: float tolerance (different summation order / FMA)
    mg.run_stage(1, n=n)
    got_out = mg.output_tensor().cpu().numpy()
    for i in range(n):
        want = oc[i].infer()
        scale = max(1.0, float(np.abs(want).max()))
        err = float(np.abs(got_out[i] - want).max()) / scale
        assert err < 1e-4, "logits rel err %g (stream %d)" % (err, i)

    # (2) decode + IIR on the ORACLE's logits, from a non-trivial previous state — bit exact
    prev = synth.random_u8((n, info["out_h"], info["out_w"]), 11)
    mg.output_tensor().copy_(_dev(np.stack([c.output() for c in oc])))
    mg.ofinal().copy_(_dev(prev))
    mg.run_stage(2, n=n)
    got_of = mg.ofinal().cpu().numpy()
    for i in range(n):
        oc[i].set_ofinal(prev[i])
        oc[i].post()
        assert np.array_equal(got_of[i], oc[i].ofinal()), "decode mismatch stream %d" % i

    # (3) upscale + blur into the persistent mask — bit exact
    mg.run_stage(3, n=n)
    got_m = mg.masks().cpu().numpy()
    for i in range(n):
        assert np.array_equal(got_m[i], oc[i].mask()), "mask mismatch stream %d: %d px" % (i, (got_m[i] != oc[i].mask()).sum())
    for c in oc:
        c.close()
    mg.close()


def test_meet_decode_edge_logits(bs, oracle):
    """Softmax-2 decode on adversarial logits: exact ties, 1-ulp and 1e-4-scale gaps (the fast path's margin), huge / infinite /
    NaN values (exp overflow → inf/inf = NaN → 255).  Must equal the oracle's expf-divide-compare byte for byte."""
    from backscrub_amd import synth
    W, H = VGA
    mg = bs.MaskGen(model_path("lite"), W, H, n_streams=2)
    info = mg.info
    oh, ow = info["out_h"], info["out_w"]
    rng = np.random.default_rng(3)
    npix = oh * ow
    base = rng.uniform(-12, 12, size=(2, npix)).astype(np.float32)
    gaps = np.concatenate([np.zeros(200), np.array([1e-7, 1e-6, 9e-5, 9.99e-5, 1e-4, 1.01e-4, 2e-4, 1e-3]).repeat(100),
                           rng.uniform(-3e-4, 3e-4, size=4000)]).astype(np.float32)
    logits = np.stack([base, base + rng.normal(0, 2, size=base.shape).astype(np.float32)], axis=-1)
    k = gaps.size
    logits[0, :k, 1] = logits[0, :k, 0] + gaps                       # near ties around the margin
    logits[0, k:k + 300, 1] = np.nextafter(logits[0, k:k + 300, 0], np.float32(np.inf))   # 1 ulp apart
    special = np.array([[90, 91], [91, 90], [1e30, 1e30], [-1e30, -1e30], [np.inf, 0], [0, np.inf], [-np.inf, -np.inf], [np.nan, 0], [0, np.nan],
                        [88.7, 88.8], [-104, -103], [-200, -100], [80, 80.0001], [-80.00001, -80], [79.9999, 80.0001]], np.float32)
    logits[1, :special.shape[0]] = special
    logits = logits.reshape(2, oh, ow, 2)
    prev = synth.random_u8((2, oh, ow), 9)
    mg.output_tensor().copy_(_dev(logits))
    mg.ofinal().copy_(_dev(prev))
    mg.run_stage(2, n=2)
    got = mg.ofinal().cpu().numpy()
    for i in range(2):
        want = oracle.decode_iir(info["model_type"], logits[i], prev[i])
        assert np.array_equal(got[i], want), "stream %d: %d px differ" % (i, (got[i] != want).sum())
    mg.close()


def test_mlkit_decode_edge_probabilities(bs, oracle):
    """MLKit decode (lib/libbackscrub.cc:333-341) on adversarial probabilities: `p > 0.65` compares the f32 value PROMOTED TO DOUBLE with the double literal 0.65, so
    the two floats either side of 0.65 (0.64999998 → 255, 0.65000004 → 0), values a few ulps away, 0 / 1 / beyond, subnormals, infinities and NaN (compare false →
    255) must all decode like the oracle, over a random previous state of the temporal filter."""
    from backscrub_amd import synth
    W, H = VGA
    mg = bs.MaskGen(model_path("mlkit"), W, H, n_streams=2)
    info = mg.info
    oh, ow = info["out_h"], info["out_w"]
    assert info["out_c"] == 1 if "out_c" in info else True
    rng = np.random.default_rng(11)
    p = rng.uniform(0, 1, size=(2, oh * ow)).astype(np.float32)
    below = np.float32(0.65)                                           # the f32 nearest to 0.65 lies BELOW the double 0.65
    assert float(below) < 0.65 < float(np.nextafter(below, np.float32(1)))
    edge = [below]
    for _ in range(8):
        edge.append(np.nextafter(edge[-1], np.float32(1)))
    for _ in range(8):
        edge.insert(0, np.nextafter(edge[0], np.float32(0)))
    special = np.array(edge + [0.0, -0.0, 1.0, 1.0000001, -1.0, 0.5, 0.6499999, 0.6500001, 1e-45, -1e-45, 3e38, -3e38, np.inf, -np.inf, np.nan, -np.nan], np.float32)
    p[0, :special.size] = special
    p[1, :4000] = (0.65 + rng.uniform(-3e-6, 3e-6, size=4000)).astype(np.float32)
    logits = p.reshape(2, oh, ow, 1)
    prev = synth.random_u8((2, oh, ow), 13)
    mg.output_tensor().copy_(_dev(logits.reshape(tuple(mg.output_tensor().shape))))
    mg.ofinal().copy_(_dev(prev))
    mg.run_stage(2, n=2)
    got = mg.ofinal().cpu().numpy()
    for i in range(2):
        want = oracle.decode_iir(info["model_type"], logits[i], prev[i])
        assert np.array_equal(got[i], want), "stream %d: %d px differ" % (i, (got[i] != want).sum())
    vals = oracle.decode_iir(info["model_type"], logits[0], np.zeros_like(prev[0])).reshape(-1)
    assert vals[8] == 0xE0 and vals[9] == 0x00                          # 0.64999998 is NOT > 0.65 (background, 255 & 0xE0); 0.65000004 is (person, 0)
    mg.close()


def test_deeplab_decode_edge_logits(bs, oracle):
    """DeepLab decode (lib/libbackscrub.cc:318-332): 21-way argmax with `maxval` starting at -10000 and a strict `>` (the FIRST maximum wins; a pixel whose
    logits are all <= -10000 keeps class 0), person = class 15.  Adversarial rows: exact ties between person and an earlier / a later class, 1-ulp gaps, everything
    below the initial maximum, +-inf, NaN in and around the person channel (NaN never compares greater)."""
    from backscrub_amd import synth
    W, H = VGA
    mg = bs.MaskGen(model_path("deeplab"), W, H, n_streams=2)
    info = mg.info
    oh, ow = info["out_h"], info["out_w"]
    t = mg.output_tensor()
    if t.numel() != 2 * oh * ow * 21:
        mg.close()
        pytest.skip("this build keeps no full-resolution logits tensor (the argmax tail reads the 33x33 tensor): %s" % (tuple(t.shape),))
    rng = np.random.default_rng(17)
    lg = rng.normal(0, 4, size=(2, oh * ow, 21)).astype(np.float32)
    rows = []
    def row(**kw):
        r = np.full(21, -3.0, np.float32)
        for k_, v_ in kw.items():
            r[int(k_[1:])] = v_
        rows.append(r)
    row(c15=5.0, c3=5.0)                  # tie with an EARLIER class: class 3 wins → background
    row(c15=5.0, c18=5.0)                 # tie with a LATER class: person wins
    row(c15=5.0, c3=np.nextafter(np.float32(5.0), np.float32(9)))
    row(c15=np.nextafter(np.float32(5.0), np.float32(9)), c3=5.0)
    rows.append(np.full(21, -10000.0, np.float32))                 # nothing is > -10000: class 0
    rows.append(np.full(21, -20000.0, np.float32))
    r = np.full(21, -20000.0, np.float32); r[15] = -9999.0; rows.append(r)     # only person clears the initial maximum
    r = np.full(21, -20000.0, np.float32); r[15] = -10000.0; rows.append(r)    # ... and exactly at it: not greater
    row(c15=np.inf); row(c15=np.inf, c2=np.inf); row(c15=-np.inf); row(c15=np.nan); row(c15=9.0, c14=np.nan); row(c15=9.0, c0=np.nan); row(c0=np.nan, c15=np.nan)
    rows.append(np.full(21, np.nan, np.float32)); rows.append(np.full(21, np.inf, np.float32)); rows.append(np.zeros(21, np.float32))
    lg[0, :len(rows)] = np.stack(rows)
    k = 3000                                                          # many near ties between person and a random other class
    other = rng.integers(0, 21, size=k)
    lg[1, np.arange(k), 15] = 7.0
    lg[1, np.arange(k), other] = (7.0 + rng.choice([0.0, 1e-6, -1e-6, 5e-7], size=k)).astype(np.float32)
    logits = lg.reshape(2, oh, ow, 21)
    prev = synth.random_u8((2, oh, ow), 19)
    t.copy_(_dev(logits.reshape(tuple(t.shape))))
    mg.ofinal().copy_(_dev(prev))
    mg.run_stage(2, n=2)
    got = mg.ofinal().cpu().numpy()
    for i in range(2):
        want = oracle.decode_iir(info["model_type"], logits[i], prev[i])
        assert np.array_equal(got[i], want), "stream %d: %d px differ" % (i, (got[i] != want).sum())
    v = oracle.decode_iir(info["model_type"], logits[0], np.zeros_like(prev[0])).reshape(-1)
    assert (v[0], v[1], v[4], v[6], v[7]) == (0xE0, 0x00, 0xE0, 0x00, 0xE0)
    mg.close()


@pytest.mark.parametrize("res", [VGA, (322, 242)])
def test_generic_mask_kernel_matches_tile_kernel(bs, oracle, res, monkeypatch, debug_switches):
    """The single-round-trip tile kernel (default) and the generic mask kernel (BSX_NO_MASK_TILE=1; also the fallback when
    a tile's source block does not fit LDS) must both be bit-exact against the oracle, stand-alone and fused with the blend."""
    from backscrub_amd import synth
    W, H = res
    path = model_path("lite")
    n = 3
    frames = np.stack([synth.frame(W, H, s) for s in range(n)])
    bg = synth.background(W, H)
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("BSX_NO_MASK_TILE", "1")
        else:
            monkeypatch.delenv("BSX_NO_MASK_TILE", raising=False)
        # stand-alone: decode + upscale + blur from the oracle's logits over a noisy previous state (every weight exercised)
        mg = bs.MaskGen(path, W, H, n_streams=n)
        oc = [oracle.Ctx(path, W, H) for _ in range(n)]
        info = mg.info
        prev = synth.random_u8((n, info["out_h"], info["out_w"]), 5)
        for i in range(n):
            oc[i].prep(frames[i])
            oc[i].infer()
            oc[i].set_ofinal(prev[i])
        mg.output_tensor().copy_(_dev(np.stack([c.output() for c in oc])))
        mg.ofinal().copy_(_dev(prev))
        mg.run_stage(2, n=n)
        mg.run_stage(3, n=n)
        got = mg.masks().cpu().numpy()
        for i in range(n):
            oc[i].post()
            assert np.array_equal(got[i], oc[i].mask()), "generic=%s stream %d: %d px differ" % (generic, i, (got[i] != oc[i].mask()).sum())
        for c in oc:
            c.close()
        mg.close()
        # fused with the composite (bsx_step_batch)
        mg = bs.MaskGen(path, W, H, n_streams=n)
        oc = [oracle.Ctx(path, W, H) for _ in range(n)]
        out = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
        mg.step(_dev(frames), _dev(bg), out)
        got_m, got_o, got_of = mg.masks().cpu().numpy(), out.cpu().numpy(), mg.ofinal().cpu().numpy()
        for i in range(n):
            want = oc[i].process(frames[i])
            assert _iou_fg(got_m[i], want) >= 0.999
            if np.array_equal(got_of[i], oc[i].ofinal()):    # same model-resolution mask → everything after it is integer work
                assert np.array_equal(got_m[i], want), "fused, generic=%s stream %d" % (generic, i)
                assert np.array_equal(got_o[i], oracle.alpha_blend(bg, frames[i], want))
        for c in oc:
            c.close()
        mg.close()
    monkeypatch.delenv("BSX_NO_MASK_TILE", raising=False)


NETWORK_PATHS = {
    # the Meet / MLKit graphs have three executions of the same fused plan; all must meet the logit tolerance against the oracle
    # (the middle program itself in two forms: the graph-specialised kernel compiled by hipRTC at bsx_new — the default — and the micro-op interpreter)
    "lite": [("segments + specialised middle kernel (hipRTC)", {}, "program execution: specialised kernel"),
             ("segments + interpreted middle program", {"BSX_NO_RTC": "1"}, "program execution: interpreted (BSX_NO_RTC)"),
             ("whole-network program", {"BSX_NO_SEGMENTS": "1"}, "frame program: ON"),
             ("one launch per step", {"BSX_NO_FRAME_PROGRAM": "1"}, "frame program: off")],
    "mlkit": [("segments + specialised middle kernel (hipRTC)", {}, "program execution: specialised kernel"),
              ("segments + interpreted middle program", {"BSX_NO_RTC": "1"}, "program execution: interpreted (BSX_NO_RTC)"),
              ("whole-network program", {"BSX_NO_SEGMENTS": "1"}, "frame program: ON"),
              ("one launch per step", {"BSX_NO_FRAME_PROGRAM": "1"}, "frame program: off")],
    # DeepLab runs per launch: the split-f16 MFMA GEMM (default) and the f32 MFMA GEMM, with and without the planner's rewrites
    "deeplab": [("split-f16 MFMA GEMM, fused head and expand+depthwise kernels", {}, "fused with step"),
                ("f32 MFMA GEMM", {"BSX_F16_GEMM": "off"}, "conv#66-pool"),
                ("no graph rewrites", {"BSX_NO_REWRITES": "1"}, "concat#65"),
                ("one launch per layer (no fused head, no fused expand+depthwise)", {"BSX_NO_IR_FUSE": "1", "BSX_NO_HEAD0": "1"}, "conv#66-pool")],
}


@pytest.mark.parametrize("key", ["lite", "mlkit", "deeplab"])
def test_every_execution_path_of_the_network_agrees_with_the_oracle(bs, oracle, key, monkeypatch, debug_switches):
    """Same plan, different executions (segment kernels + per-frame middle program / whole-network program / one launch per
    step; f32 vs split-f16 MFMA GEMMs; with and without the linear-algebra rewrites): every one within 1e-4 of the oracle's logits."""
    from backscrub_amd import synth
    path = model_path(key)
    W, H = VGA
    f = synth.frame(W, H, 5)
    oc = oracle.Ctx(path, W, H)
    oc.prep(f)
    want = oc.infer()
    knobs = ("BSX_F32_INPUT", "BSX_ACT16", "BSX_NO_RTC", "BSX_NO_SEGMENTS", "BSX_NO_FRAME_PROGRAM", "BSX_F16_GEMM", "BSX_NO_REWRITES", "BSX_FORCE_FRAME_PROGRAM", "BSX_NO_IR_FUSE", "BSX_NO_HEAD0")
    for name, env, marker in NETWORK_PATHS[key]:
        for k in knobs:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        mg = bs.MaskGen(path, W, H, n_streams=2)
        assert marker in mg.plan(), "%s: expected %r in the plan" % (name, marker)
        mg.run_stage(0, _dev(np.stack([f, f])))
        mg.run_stage(1, n=2)
        got = mg.output_tensor().cpu().numpy()
        assert np.array_equal(got[0], got[1])
        err = float(np.abs(got[0] - want).max()) / max(1.0, float(np.abs(want).max()))
        assert err < 1e-4, "%s: rel err %g" % (name, err)
        mg.close()
    for k in knobs:
        monkeypatch.delenv(k, raising=False)
    oc.close()


def test_deeplab_batch_of_eight_uses_the_gemm_kernels_on_every_level(bs, oracle):
    """With 8 streams even the 33x33 layers have M >= 8192 rows, i.e. every pointwise convolution runs as the split-f16 MFMA GEMM (64-, 80- and
    48-column tiles) next to the fused kernels: logits of all eight streams within 1e-4 of the oracle's; 8 x 1089 rows = 68 full 128-row blocks + a block of 8 rows."""
    from backscrub_amd import synth
    path = model_path("deeplab")
    W, H = VGA
    n = 8
    frames = np.stack([synth.frame(W, H, i % 3, i) for i in range(n)])
    mg = bs.MaskGen(path, W, H, n_streams=n)
    mg.run_stage(0, _dev(frames))
    mg.run_stage(1, n=n)
    got = mg.output_tensor().cpu().numpy()
    oc = oracle.Ctx(path, W, H)
    for i in (0, 3, 7):
        oc.prep(frames[i])
        want = oc.infer()
        err = float(np.abs(got[i] - want).max()) / max(1.0, float(np.abs(want).max()))
        assert err < 1e-4, "stream %d: rel err %g" % (i, err)
    oc.close()
    mg.close()


def test_deeplab_chained_head_equals_the_three_gemms(bs, monkeypatch, debug_switches):
    """From 8192 pixels up the ASPP head (160 -> 256 -> 256 + per-frame pooled bias -> 21) runs as ONE kernel whose 256-channel tensors stay in registers
    (pw_chain3_k); the planner's BSX_NO_CHAIN3 (debug build) keeps the three split-f16 GEMM launches.  Same operand split, same term order, K slabs ascending:
    the logits differ by rounding inside one MFMA only.  9 streams of DIFFERENT frames: 9801 pixels = 76 full 128-pixel workgroups + one of 73 pixels, and
    16-pixel tiles that straddle two frames (two per-frame bias vectors inside one tile)."""
    from backscrub_amd import synth
    path = model_path("deeplab")
    W, H = VGA
    n = 9
    frames = np.stack([synth.frame(W, H, i % 3, 11 + i) for i in range(n)])
    outs = {}
    for label, env in (("chained", None), ("three launches", "1")):
        monkeypatch.delenv("BSX_NO_CHAIN3", raising=False)
        if env:
            monkeypatch.setenv("BSX_NO_CHAIN3", env)
        mg = bs.MaskGen(path, W, H, n_streams=n)
        assert ("chained with steps" in mg.plan()) == (env is None), label
        mg.run_stage(0, _dev(frames))
        mg.run_stage(1, n=n)
        outs[label] = mg.output_tensor().cpu().numpy().copy()
        mg.close()
    monkeypatch.delenv("BSX_NO_CHAIN3", raising=False)
    a, b = outs["chained"], outs["three launches"]
    assert np.isfinite(a).all()
    err = float(np.abs(a - b).max()) / max(1.0, float(np.abs(b).max()))
    assert err < 2e-6, "chained vs three launches: rel err %g" % err
    assert not np.array_equal(a[0], a[1])                      # different frames: the per-frame bias matters
    assert (a.argmax(-1) == b.argmax(-1)).mean() > 0.9999


def test_deeplab_f16_storage_mode_is_gated_by_iou(bs, oracle, monkeypatch):
    """BSX_F16_GEMM=fast16: plain f16 MFMA operands AND the depthwise outputs of the fused blocks stored as f16 (half the traffic of the largest
    tensors that still reach HBM).  Opt-in, IoU-gated like `fast`: the mask of the photo fixture stays within IoU 0.995 of the oracle's and the
    logits within 2e-2, and the mode must not be the default."""
    from backscrub_amd import synth
    path = model_path("deeplab")
    W, H = VGA
    n = 8                                                             # >= 8 streams: every layer takes the GEMM / f16-input form
    frames = np.stack([synth.frame(W, H, i % 3, i) for i in range(n)])
    monkeypatch.setenv("BSX_F16_GEMM", "fast16")
    mg = bs.MaskGen(path, W, H, n_streams=n)
    monkeypatch.delenv("BSX_F16_GEMM")
    mg.run_stage(0, _dev(frames))
    mg.run_stage(1, n=n)
    got = mg.output_tensor().cpu().numpy()
    oc = oracle.Ctx(path, W, H)
    errs = []
    for i in (0, 5):
        oc.prep(frames[i])
        want = oc.infer()
        errs.append(float(np.abs(got[i] - want).max()) / max(1.0, float(np.abs(want).max())))
        agree = (got[i].argmax(-1) == want.argmax(-1)).mean()
        assert agree >= 0.995, "stream %d: argmax agreement %.5f" % (i, agree)
    assert max(errs) < 2e-2 and max(errs) > 1e-5, errs                # close, and visibly NOT the f32-grade default
    oc.close()
    mg.close()


@pytest.mark.parametrize("key,res", [("lite", (322, 242)), ("deeplab", (641, 479)), ("mlkit", (322, 242))])
def test_prep_on_odd_frame_sizes_matches_the_oracle(bs, oracle, key, res):
    """prep_fused_k (resize of the tile + halo into LDS, bilateral from LDS) on geometries whose tiles do not divide the canvas and whose rows are not dword
    multiples, incl. a stream of pure noise: the network input bit for bit the oracle's.  (Round 6 deleted the two-launch form through a stored canvas this kernel
    used to be cross-checked against; the oracle is the check, as in test_stages_match_oracle.)"""
    from backscrub_amd import synth
    path = model_path(key)
    W, H = res
    frames = np.stack([synth.frame(W, H, 0), synth.frame(W, H, 1, 3), synth.random_u8((H, W, 3), 7)])
    mg = bs.MaskGen(path, W, H, n_streams=4)
    oc = oracle.Ctx(path, W, H)
    mg.run_stage(0, _dev(frames))
    got = mg.input_tensor()[:3].cpu().numpy()
    for i in range(3):
        want = oc.prep(frames[i])
        assert np.array_equal(got[i], want), "stream %d: %d values differ" % (i, (got[i] != want).sum())
    oc.close()
    mg.close()


@pytest.mark.parametrize("key,res", [("lite", VGA), ("mlkit", HD), ("full", HD), ("deeplab", VGA)])
def test_stems_that_read_the_8bit_input_are_bit_identical_to_the_f32_tensor(bs, monkeypatch, key, res, debug_switches):
    """The step's prep hands the stem the filtered 8-bit pixels and the stem applies convertTo's two roundings while it stages its input window
    (seg_head_k<.., U8IN>, dl_head0_k<true>): the 12 B/px f32 input tensor of libbackscrub.cc:302 is never written.  Same logits, bit for bit, as
    the form that reads the f32 tensor (BSX_F32_INPUT=1) — incl. a stream of pure noise and a partial batch."""
    from backscrub_amd import synth
    path = model_path(key)
    W, H = res
    n = 5
    rng = np.random.default_rng(3)
    frames = np.stack([synth.frame(W, H, i, 2 * i) for i in range(n - 1)] + [rng.integers(0, 256, (H, W, 3), dtype=np.uint8)])
    outs = []
    for f32 in (False, True):
        if f32:
            monkeypatch.setenv("BSX_F32_INPUT", "1")
        mg = bs.MaskGen(path, W, H, n_streams=8)
        monkeypatch.delenv("BSX_F32_INPUT", raising=False)
        mg.run_stage(0, _dev(frames))
        mg.run_stage(1, n=n)
        outs.append((mg.input_tensor()[:n].clone(), mg.output_tensor()[:n].clone()))
        mg.close()
    assert torch.equal(outs[0][0], outs[1][0])            # the stage entry materialises the f32 tensor in both forms (what stage-0 tests read)
    assert torch.equal(outs[0][1], outs[1][1]), "logits differ between the 8-bit and the f32 network input"


@pytest.mark.parametrize("key,res", [("lite", VGA), ("mlkit", HD), ("full", HD)])
def test_act16_storage_mode_of_the_segmented_networks_is_gated(bs, oracle, monkeypatch, key, res):
    """BSX_ACT16=1 (g1 for Meet / MLKit): every activation tensor that crosses a kernel boundary or spills out of LDS is STORED as f16
    (A, b0, B, c0, lo2, lo and the 12x20 / 16x16 level tensors of the middle program); arithmetic stays f32.  Opt-in and gated like the
    DeepLab modes: logits within 2e-2 of the oracle, decisions agree on >= 99.5 % of the model pixels, end-to-end IoU >= 0.99 on the photo
    fixture — and visibly NOT the f32 default, which must stay the default."""
    from backscrub_amd import synth
    from tools import make_photo_fixture as P
    path = model_path(key)
    W, H = res
    n = 3
    frames = np.stack([synth.frame(W, H, i, i) for i in range(n)])
    monkeypatch.setenv("BSX_ACT16", "1")
    mg = bs.MaskGen(path, W, H, n_streams=n)
    monkeypatch.delenv("BSX_ACT16")
    assert "16-bit activation storage" in mg.plan()
    mg.run_stage(0, _dev(frames))
    mg.run_stage(1, n=n)
    got = mg.output_tensor().cpu().numpy()
    oc = oracle.Ctx(path, W, H)
    errs = []
    for i in range(n):
        oc.prep(frames[i])
        want = oc.infer()
        errs.append(float(np.abs(got[i] - want).max()) / max(1.0, float(np.abs(want).max())))
        dec = (lambda t: t[..., 1] > t[..., 0]) if want.shape[-1] == 2 else (lambda t: t[..., 0] > 0.65)
        agree = (dec(got[i]) == dec(want)).mean()
        assert agree >= 0.995, "stream %d: decisions agree on %.5f" % (i, agree)
    assert max(errs) < 2e-2 and max(errs) > 1e-5, errs                # close, and visibly not the f32 default
    oc.close()
    mg.close()
    default = bs.MaskGen(path, W, H, n_streams=1)
    assert "16-bit activation storage" not in default.plan()
    default.close()
    if "synthetic" in os.path.basename(path) or res != VGA:
        return
    # end to end on the photo fixture (real weights, a real person in the frame)
    photo = P.load_frames()
    monkeypatch.setenv("BSX_ACT16", "1")
    mg = bs.MaskGen(path, W, H, n_streams=photo.shape[0])
    monkeypatch.delenv("BSX_ACT16")
    ocs = [oracle.Ctx(path, W, H) for _ in range(photo.shape[0])]
    bg = synth.background(W, H)
    out = torch.empty((photo.shape[0], H, W, 3), dtype=torch.uint8, device="cuda")
    for t in range(3):
        mg.step(_dev(photo), _dev(bg), out)
        masks = mg.masks().cpu().numpy()
        for i, c in enumerate(ocs):
            iou = _iou_fg(masks[i], c.process(photo[i]), need_person=(t == 2))
            assert iou >= 0.99, "%s t=%d frame %d IoU %.5f" % (key, t, i, iou)
    for c in ocs:
        c.close()
    mg.close()


def test_deeplab_fast_f16_mode_is_close_but_not_parity_grade(bs, oracle, monkeypatch):
    """BSX_F16_GEMM=fast (plain f16 MFMA inputs, f32 accumulate — what SetAllowFp16PrecisionForFp32 permits, lib/libbackscrub.cc:225)
    is an opt-in mode: its logits are close (1e-2) but it is NOT held to the 1e-4 parity bar; the default split-f16 mode is."""
    from backscrub_amd import synth
    path = model_path("deeplab")
    W, H = VGA
    f = synth.frame(W, H, 5)
    oc = oracle.Ctx(path, W, H)
    oc.prep(f)
    want = oc.infer()
    monkeypatch.setenv("BSX_F16_GEMM", "fast")
    mg = bs.MaskGen(path, W, H, n_streams=1)
    mg.run_stage(0, _dev(f[None]))
    mg.run_stage(1, n=1)
    got = mg.output_tensor().cpu().numpy()[0]
    err = float(np.abs(got - want).max()) / max(1.0, float(np.abs(want).max()))
    assert err < 1e-2, err
    assert (got.argmax(-1) == want.argmax(-1)).mean() > 0.99
    mg.close(); oc.close()
    monkeypatch.delenv("BSX_F16_GEMM", raising=False)


# --------------------------------------------------------------------------------------------
# end to end: several frames per stream, temporal state carried on the GPU
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("key,res", [("lite", VGA), ("mlkit", HD), ("full", HD), ("deeplab", VGA)])
@pytest.mark.parametrize("real", [True, False])
def test_end_to_end_iou_and_composite(bs, oracle, key, res, real):
    from backscrub_amd import synth
    path = model_path(key, prefer_real=real)
    if real and "synthetic" in os.path.basename(path):
        pytest.skip("reference model not staged on this box")
    W, H = res
    n, T = (4, 4) if key != "deeplab" else (2, 3)
    mg = bs.MaskGen(path, W, H, n_streams=n)
    oc = [oracle.Ctx(path, W, H) for _ in range(n)]
    bg = synth.background(W, H)
    d_bg = _dev(bg)
    out = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    stats = {"frames": 0, "frames_with_identical_masks": 0, "mask_pixels_differing": 0, "composite_pixels_over_1_lsb": 0, "composite_max_abs": 0}
    for t in range(T):
        frames = np.stack([synth.frame(W, H, s, t) for s in range(n)])
        d_frames = _dev(frames)
        mg.step(d_frames, d_bg, out)
        got_mask = mg.masks().cpu().numpy()
        got_out = out.cpu().numpy()
        for i in range(n):
            want_mask = oc[i].process(frames[i])
            want_out = oracle.alpha_blend(bg, frames[i], want_mask)
            iou = _iou_fg(got_mask[i], want_mask)
            assert iou >= 0.999, "t=%d stream %d IoU %.5f" % (t, i, iou)
            if np.array_equal(got_mask[i], want_mask):
                assert np.array_equal(got_out[i], want_out)
            # composite: the blend is bit-exact, so the composite can only differ where the full-resolution masks differ (a flipped decision
            # pixel of the model-resolution mask, bounded by the IoU bar) — everywhere else the difference is 0, not just <= 1 LSB
            diff = np.abs(got_out[i].astype(np.int16) - want_out.astype(np.int16)).max(-1)
            same = got_mask[i] == want_mask
            assert int(diff[same].max(initial=0)) == 0, "t=%d stream %d: composite differs where the masks agree" % (t, i)
            assert (~same).mean() <= 1e-3, "t=%d stream %d: %.5f of mask pixels differ" % (t, i, (~same).mean())
            # the north star's literal bar (max-abs <= 1 LSB): holds on the WHOLE frame whenever the masks agree (the usual case — counted below), and the pixels
            # above 1 LSB are a subset of the (counted, bounded) mask pixels that differ
            over = diff > 1
            assert not np.any(over & same) and int(over.sum()) <= int((~same).sum())
            if real:               # the reference's weights on a BASELINE geometry: the north star's literal bar, every frame, the whole frame
                assert int(diff.max()) <= 1, "t=%d stream %d: composite max-abs %d (%d pixels > 1 LSB)" % (t, i, int(diff.max()), int(over.sum()))
            stats["frames"] += 1
            stats["frames_with_identical_masks"] += int(same.all())
            stats["mask_pixels_differing"] += int((~same).sum())
            stats["composite_pixels_over_1_lsb"] += int(over.sum())
            stats["composite_max_abs"] = max(stats["composite_max_abs"], int(diff.max()))
    print("end-to-end %s %dx%d %s weights: %s" % (key, W, H, "real" if real else "synthetic", stats))
    if real:                       # the four BASELINE geometries with the reference's weights: the literal bar held on every frame above
        assert stats["composite_max_abs"] <= 1 and stats["composite_pixels_over_1_lsb"] == 0
    for c in oc:
        c.close()
    mg.close()


@pytest.mark.parametrize("key", ["deeplab", "lite", "full", "mlkit"])
def test_end_to_end_on_the_photo_fixture(bs, oracle, key):
    """End to end on REAL pixels with the reference's REAL weights: the two 640x480 webcam screenshots of the reference's
    backgrounds/screenshot.jpg (tests/golden/photo_2x640x480.png).  All four networks segment the person (~24 % of the frame),
    so the IoU here is over a real person region — in particular for DeepLab, which finds nobody in the synthetic frames."""
    from backscrub_amd import synth
    from tools import make_photo_fixture as P
    path = model_path(key)
    if "synthetic" in os.path.basename(path):
        pytest.skip("reference model not staged on this box")
    W, H = VGA
    frames = P.load_frames()
    n = frames.shape[0]
    mg = bs.MaskGen(path, W, H, n_streams=n)
    oc = [oracle.Ctx(path, W, H) for _ in range(n)]
    bg = synth.background(W, H)
    out = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    for t in range(3):
        mg.step(_dev(frames), _dev(bg), out)
        got_mask, got_out = mg.masks().cpu().numpy(), out.cpu().numpy()
        for i in range(n):
            want_mask = oc[i].process(frames[i])
            iou = _iou_fg(got_mask[i], want_mask, need_person=(t == 2))
            assert iou >= 0.999, "%s t=%d frame %d IoU %.5f" % (key, t, i, iou)
            want_out = oracle.alpha_blend(bg, frames[i], want_mask)
            # the north star's bar, as stated: max-abs <= 1 LSB over the whole composited frame (default, parity-grade path, real weights)
            diff = np.abs(got_out[i].astype(np.int16) - want_out.astype(np.int16)).max(-1)
            assert int(diff.max()) <= 1, "%s t=%d frame %d: composite max-abs %d (%d pixels > 1 LSB)" % (key, t, i, int(diff.max()), int((diff > 1).sum()))
    for c in oc:
        c.close()
    mg.close()


def test_deeplab_argmax_agreement(bs, oracle):
    """DeepLab rarely fires 'person' on synthetic frames, so also compare the full 21-way argmax map."""
    from backscrub_amd import synth
    path = model_path("deeplab")
    W, H = VGA
    mg = bs.MaskGen(path, W, H, n_streams=1)
    oc = oracle.Ctx(path, W, H)
    f = synth.frame(W, H, 3)
    mg.process_batch(_dev(f[None]))
    got = mg.output_tensor().cpu().numpy()[0].argmax(-1)
    oc.process(f)
    want = oc.output().argmax(-1)
    assert (got == want).mean() >= 0.999
    oc.close()
    mg.close()


def test_deeplab_tail_forms_agree(bs, monkeypatch, debug_switches):
    """The fused resize + argmax + IIR tail has two forms: 24-float LDS pixels read as quads with the person test from two running maxima (21
    classes), and the scalar first-maximum scan over 32-float pixels (what more than 24 classes would take, BSX_TAIL_GENERIC=1).  Same
    interpolation arithmetic, same decision: on the photo fixture (a real person: both byte values occur) the decoded `ofinal` bytes after three
    steps (the IIR carries over) must be identical."""
    from tools import make_photo_fixture as P
    path = model_path("deeplab")
    if "synthetic" in os.path.basename(path):
        pytest.skip("reference model not staged on this box")
    W, H = VGA
    frames = P.load_frames()
    n = frames.shape[0]
    outs = []
    for generic in (False, True):
        if generic:
            monkeypatch.setenv("BSX_TAIL_GENERIC", "1")
        mg = bs.MaskGen(path, W, H, n_streams=n)
        for t in range(3):
            mg.process_batch(_dev(frames if t != 1 else frames[::-1].copy()))
        outs.append(mg.ofinal().cpu().numpy().copy())
        mg.close()
    monkeypatch.delenv("BSX_TAIL_GENERIC")
    assert outs[0].shape == outs[1].shape and outs[0].size > 0
    assert np.array_equal(outs[0], outs[1])
    assert len(np.unique(outs[0])) > 2            # person and background pixels at several IIR levels


# --------------------------------------------------------------------------------------------
# the drop-in single-frame host path and its reference-style error behaviour
# --------------------------------------------------------------------------------------------
def test_bs_maskgen_host_path_and_callbacks(bs, oracle):
    from backscrub_amd import synth
    path = model_path("lite")
    W, H = VGA
    events = []
    ctx = bs.bs_maskgen_new(path, 2, W, H, None, lambda c: events.append("prep"), lambda c: events.append("infer"),
                            lambda c: events.append("mask"), None)
    assert ctx is not None
    oc = oracle.Ctx(path, W, H)
    mask = np.zeros((H, W), np.uint8)
    for t in range(3):
        f = synth.frame(W, H, 0, t)
        assert bs.bs_maskgen_process(ctx, f, mask) is True
        assert _iou_fg(mask, oc.process(f)) >= 0.999
    assert events == ["prep", "infer", "mask"] * 3   # lib/libbackscrub.cc:303,311,363
    bs.bs_maskgen_delete(ctx)
    bs.bs_maskgen_delete(None)                        # NULL-safe, :262
    assert bs.bs_maskgen_process(None, f, mask) is False  # :280
    msgs = []
    assert bs.bs_maskgen_new("/nonexistent/segm_x.tflite", 2, W, H, lambda c, m: msgs.append(m)) is None  # :191-195
    assert msgs and b"unable to load model" in msgs[0]
    assert bs.bs_maskgen_new(synthetic_model_path("lite").replace("segm_", "xx_") + ".missing", 2, W, H, lambda c, m: None) is None
    oc.close()


def test_unknown_model_type_rejected(bs, tmp_path):
    import shutil
    p = tmp_path / "mystery.tflite"
    shutil.copy(synthetic_model_path("lite"), p)
    msgs = []
    assert bs.bs_maskgen_new(str(p), 2, 640, 480, lambda c, m: msgs.append(m)) is None   # :199-203
    assert b"unknown model type" in msgs[0]


def test_roi_border_stays_background(bs):
    """Mask pixels outside roidim stay 255 forever (libbackscrub.cc:248-249)."""
    from backscrub_amd import synth
    W, H = VGA
    mg = bs.MaskGen(model_path("mlkit"), W, H, n_streams=1)   # roi = (80,0,480,480)
    for t in range(3):
        m = mg.process_batch(_dev(synth.frame(W, H, 0, t)[None])).cpu().numpy()[0]
    x, _, w, _ = mg.info["roi"]
    assert (m[:, :x] == 255).all() and (m[:, x + w:] == 255).all()
    mg.close()


# --------------------------------------------------------------------------------------------
# "next" rows already on the GPU: background resize and the YUYV packer — bit exact
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("src,dst", [((1280, 720), (640, 480)), ((1200, 859), (640, 480)), ((480, 360), (1280, 720)),
                                     ((1280, 960), (640, 480)), ((640, 480), (640, 480)), ((33, 17), (640, 480))])
def test_resize_bgr_matches_oracle(bs, oracle, src, dst):
    from backscrub_amd import synth
    img = synth.random_u8((2, src[1], src[0], 3), 5)
    mg = bs.MaskGen(synthetic_model_path("lite"), 640, 480, n_streams=1)
    got = mg.resize_bgr(_dev(img), dst[0], dst[1]).cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], oracle.resize_linear(img[i], dst[0], dst[1]))
    mg.close()


def test_yuyv_matches_oracle(bs, oracle):
    from backscrub_amd import synth
    img = synth.random_u8((2, 480, 640, 3), 9)
    mg = bs.MaskGen(synthetic_model_path("lite"), 640, 480, n_streams=1)
    got = mg.bgr_to_yuyv(_dev(img)).cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], oracle.bgr_to_yuyv(img[i]))
    mg.close()


@pytest.mark.parametrize("key,res", [("lite", VGA), ("mlkit", VGA), ("full", (1280, 720)), ("lite", (322, 242))])
def test_step_with_fused_yuyv_output(bs, oracle, key, res):
    """bsx_step_batch_yuyv = alpha_blend then convert_rgb_to_yuyv (app/deepseg.cc:661,681) in one pass: bit-identical to the two-call
    form and to the oracle's packer on the composite — full-frame ROI (lite/VGA), ROI with background strips outside (mlkit/VGA), HD,
    and a geometry the fused tile kernel does not take (322x242: scratch + packer fallback)."""
    import torch
    from backscrub_amd import synth
    W, H = res
    n = 3
    mg_a = bs.MaskGen(model_path(key), W, H, n_streams=n)
    mg_b = bs.MaskGen(model_path(key), W, H, n_streams=n)
    bg = _dev(synth.random_u8((n, H, W, 3), 77))
    out = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    yuyv = torch.empty((n, H, W, 2), dtype=torch.uint8, device="cuda")
    for t in range(3):
        frames = _dev(np.stack([synth.frame(W, H, i, t) for i in range(n)]))
        mg_a.step(frames, bg, out)
        mg_b.step_yuyv(frames, bg, yuyv)
        want = mg_a.bgr_to_yuyv(out).cpu().numpy()
        got = yuyv.cpu().numpy()
        assert np.array_equal(got, want), "t=%d: %d bytes differ" % (t, (got != want).sum())
        comp = out.cpu().numpy()
        assert np.array_equal(got[0], oracle.bgr_to_yuyv(comp[0]))
        assert np.array_equal(mg_a.masks().cpu().numpy(), mg_b.masks().cpu().numpy())
    bg1 = bg[0].contiguous()                                             # one shared background image
    mg_a.step(frames, bg1, out)
    mg_b.step_yuyv(frames, bg1, yuyv)
    assert np.array_equal(yuyv.cpu().numpy(), mg_a.bgr_to_yuyv(out).cpu().numpy())
    mg_a.close()
    mg_b.close()


def _camera_yuyv(bs_mg, W, H, n, t, rng):
    """what a YUYV webcam would deliver for the synthetic scene: the BGR frame packed by the product's own packer is NOT it (that packer swaps U and V the way
    convert_rgb_to_yuyv does) — build Y0 U Y1 V from BT.601 directly, plus streams of pure noise (every byte value, out-of-range Y / chroma: the clamps)."""
    from backscrub_amd import synth
    fr = np.stack([synth.frame(W, H, i, t) for i in range(n)]).astype(np.float32)
    b, g, r = fr[..., 0], fr[..., 1], fr[..., 2]
    y = 16 + 0.257 * r + 0.504 * g + 0.098 * b
    u = 128 - 0.148 * r - 0.291 * g + 0.439 * b
    v = 128 + 0.439 * r - 0.368 * g - 0.071 * b
    yuyv = np.empty((n, H, W, 2), np.uint8)
    yuyv[..., 0] = np.clip(np.rint(y), 0, 255)
    yuyv[:, :, 0::2, 1] = np.clip(np.rint((u[:, :, 0::2] + u[:, :, 1::2]) / 2), 0, 255)
    yuyv[:, :, 1::2, 1] = np.clip(np.rint((v[:, :, 0::2] + v[:, :, 1::2]) / 2), 0, 255)
    yuyv[n - 1] = rng.integers(0, 256, size=(H, W, 2), dtype=np.uint8)
    return yuyv


@pytest.mark.parametrize("key,res", [("lite", VGA), ("mlkit", VGA), ("full", (1280, 720)), ("lite", (322, 242)), ("deeplab", VGA), ("mlkit", (1280, 720))])
def test_step_that_takes_the_cameras_raw_yuyv(bs, oracle, key, res):
    """BSX_STEP_YUYV_IN: cv::COLOR_YUV2BGR_YUYV (app/deepseg.cc:553,725) folded into the two kernels that read the frame.  Context B takes the
    raw 4:2:2 frames; context A gets them converted by bsx_yuyv_to_bgr (itself bit-exact against the oracle: test_yuyv_to_bgr_matches_oracle) and runs the BGR step.
    Same composite — as BGR and as YUYV out (YUYV in -> YUYV out in ONE step) —, same persistent masks, same temporal state, bit for bit, over three time steps,
    with per-stream and shared backgrounds; on the four geometries of test_step_with_fused_yuyv_output (full-frame ROI, ROI with strips outside, HD, and 322x242,
    which the fused kernels do not take: conversion into the context's scratch first) + DeepLab + MLKit/HD.  One stream is pure noise."""
    import torch
    W, H = res
    n = 3
    rng = np.random.default_rng(41)
    mg_a = bs.MaskGen(model_path(key), W, H, n_streams=n)
    mg_b = bs.MaskGen(model_path(key), W, H, n_streams=n)
    mg_c = bs.MaskGen(model_path(key), W, H, n_streams=n)
    bg = _dev(rng.integers(0, 256, size=(n, H, W, 3), dtype=np.uint8))
    out_a = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    out_b = torch.empty_like(out_a)
    y_c = torch.empty((n, H, W, 2), dtype=torch.uint8, device="cuda")
    for t in range(3):
        raw = _dev(_camera_yuyv(mg_a, W, H, n, t, rng))
        bgr = mg_a.yuyv_to_bgr(raw)
        if t == 0:
            assert np.array_equal(bgr[0].cpu().numpy(), oracle.yuyv_to_bgr(raw[0].cpu().numpy()))
            if W % 4 == 0:                                                 # the prep stage alone (stage 4 = stage 0 on raw frames): the oracle's network input
                mg_b.run_stage(4, raw)
                oc = oracle.Ctx(model_path(key), W, H)
                got_in = mg_b.input_tensor().cpu().numpy()
                for i in range(n):
                    assert np.array_equal(got_in[i], oc.prep(oracle.yuyv_to_bgr(raw[i].cpu().numpy()))), "prep on YUYV frames, stream %d" % i
                oc.close()
        b = bg if t != 1 else bg[1].contiguous()                          # per-stream backgrounds, and once ONE shared image
        mg_a.step(bgr, b, out_a)
        mg_b.step_ex(raw, b, out_b, yuyv_in=True)
        mg_c.step_ex(raw, b, y_c, yuyv=True, yuyv_in=True)
        assert torch.equal(out_b, out_a), "t=%d: %d composite bytes differ" % (t, int((out_b != out_a).sum()))
        assert torch.equal(mg_b.masks(), mg_a.masks()) and torch.equal(mg_b.ofinal(), mg_a.ofinal())
        assert torch.equal(y_c, mg_a.bgr_to_yuyv(out_a)), "t=%d: YUYV in -> YUYV out differs from convert + step + pack" % t
        assert torch.equal(mg_c.masks(), mg_a.masks()) and torch.equal(mg_c.ofinal(), mg_a.ofinal())
    # flips and composite-only, and the forms that cannot be fused (background blurred from the frame itself; output over the input buffer)
    mg_a.step_ex(bgr, bg, out_a, flip_h=True, flip_v=True, no_mask=True)
    mg_b.step_ex(raw, bg, out_b, flip_h=True, flip_v=True, no_mask=True, yuyv_in=True)
    assert torch.equal(out_b, out_a)
    if W % 4 == 0:
        mg_a.step_ex(bgr, None, out_a, bgblur=9)
        mg_b.step_ex(raw, None, out_b, bgblur=9, yuyv_in=True)
        assert torch.equal(out_b, out_a)
    buf = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    buf.view(-1)[:raw.numel()].copy_(raw.view(-1))                         # the YUYV batch at the start of the buffer the composite is written to
    mg_a.step(bgr, bg, out_a)
    mg_b.step_ex(buf.view(-1)[:raw.numel()].view(n, H, W, 2), bg, buf, yuyv_in=True)
    assert torch.equal(buf, out_a)
    with pytest.raises(bs.BsxError):
        mg_b.step_ex(bgr, bg, out_b, yuyv_in=True)                         # a BGR tensor where 4:2:2 frames are announced
    for m in (mg_a, mg_b, mg_c):
        m.close()


def test_pipelined_step_takes_yuyv_frames(bs):
    """bsx_step_batch_pipelined with BSX_STEP_YUYV_IN | BSX_STEP_YUYV: the composite of batch k — enqueued by call k + 1 on the context's own stream — reads the raw
    4:2:2 frames of batch k; results equal the synchronous YUYV in -> YUYV out step, one call later."""
    import torch
    W, H = VGA
    n, T = 4, 4
    rng = np.random.default_rng(43)
    mg_a = bs.MaskGen(model_path("lite"), W, H, n_streams=n)
    mg_b = bs.MaskGen(model_path("lite"), W, H, n_streams=n)
    bg = _dev(rng.integers(0, 256, size=(H, W, 3), dtype=np.uint8))
    raws = [_dev(_camera_yuyv(mg_a, W, H, n, t, rng)) for t in range(T)]
    want, got = [], [torch.empty((n, H, W, 2), dtype=torch.uint8, device="cuda") for _ in range(T)]
    for t in range(T):
        o = torch.empty((n, H, W, 2), dtype=torch.uint8, device="cuda")
        mg_a.step_ex(raws[t], bg, o, yuyv=True, yuyv_in=True)
        want.append(o)
        mg_b.step_pipelined(raws[t], bg, got[t], yuyv=True, yuyv_in=True)
    mg_b.flush_pipelined()
    torch.cuda.synchronize()
    for t in range(T):
        assert torch.equal(got[t], want[t]), "batch %d" % t
    assert torch.equal(mg_b.masks(), mg_a.masks()) and torch.equal(mg_b.ofinal(), mg_a.ofinal())
    mg_a.close()
    mg_b.close()


@pytest.mark.parametrize("key,res", [("lite", VGA), ("mlkit", VGA), ("full", (1280, 720)), ("lite", (322, 242))])
def test_step_with_flips_folded_into_the_blend(bs, oracle, key, res):
    """bsx_step_batch_ex: cv::flip of the composite (app/deepseg.cc:667-673) costs no pass of its own — the blend's tiles store to the mirrored
    position.  Bit-identical to step + cv::flip (numpy) [+ the oracle's YUYV packer] for all three flip codes, with and without the YUYV epilogue,
    on a full-frame ROI, a ROI with background strips outside, HD, and a geometry that takes the unfused fallback; masks stay unflipped."""
    import torch
    from backscrub_amd import synth
    W, H = res
    n = 3
    mg_a = bs.MaskGen(model_path(key), W, H, n_streams=n)
    mg_b = bs.MaskGen(model_path(key), W, H, n_streams=n)
    bg = _dev(synth.random_u8((n, H, W, 3), 78))
    out = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    got3 = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    got2 = torch.empty((n, H, W, 2), dtype=torch.uint8, device="cuda")
    t = 0
    for fh, fv in ((True, False), (False, True), (True, True)):
        for yuyv in (False, True):
            frames = _dev(np.stack([synth.frame(W, H, i, t) for i in range(n)]))
            t += 1
            mg_a.step(frames, bg, out)
            mg_b.step_ex(frames, bg, got2 if yuyv else got3, flip_h=fh, flip_v=fv, yuyv=yuyv)
            want = out.cpu().numpy()
            if fh:
                want = want[:, :, ::-1]
            if fv:
                want = want[:, ::-1]
            want = np.ascontiguousarray(want)
            if yuyv:
                want = np.stack([oracle.bgr_to_yuyv(w) for w in want])
            got = (got2 if yuyv else got3).cpu().numpy()
            assert np.array_equal(got, want), "flip_h=%s flip_v=%s yuyv=%s: %d bytes differ" % (fh, fv, yuyv, (got != want).sum())
            assert np.array_equal(mg_a.masks().cpu().numpy(), mg_b.masks().cpu().numpy())
    mg_b.step_ex(frames, bg, got3)                                        # no flags = bsx_step_batch
    mg_a.step(frames, bg, out)
    assert torch.equal(got3, out)
    mg_a.close()
    mg_b.close()


@pytest.mark.parametrize("key,res", [("lite", VGA), ("mlkit", HD), ("deeplab", VGA), ("lite", (322, 242))])
def test_uniform_tile_shortcut_is_bit_identical(bs, oracle, key, res, monkeypatch):
    """mask_tile_k skips the resize / blur phases (and the read of the operand the blend does not need) for tiles whose whole source block is 0x00 or 0xFF — the state the
    temporal filter settles into away from the person's outline.  Same bytes as the general path (BSX_NO_UNIFORM_TILES=1) and as the oracle: on hand-made states with
    uniform regions, near-uniform values (1, 254: NOT uniform), thin features and pure noise; and through five frames of the whole step (transient 0xE0 / 0xFC states,
    then the steady state), BGR / flipped / YUYV / composite-only outputs."""
    import torch
    from backscrub_amd import synth
    W, H = res
    n = 4
    path = model_path(key)
    mg_u = bs.MaskGen(path, W, H, n_streams=n)
    monkeypatch.setenv("BSX_NO_UNIFORM_TILES", "1")
    mg_g = bs.MaskGen(path, W, H, n_streams=n)
    monkeypatch.delenv("BSX_NO_UNIFORM_TILES")
    i = mg_u.info
    oh, ow = i["out_h"], i["out_w"]
    rng = np.random.default_rng(5)
    states = np.zeros((n, oh, ow), np.uint8)
    states[0, : oh // 2] = 255                                     # two uniform halves, one horizontal edge
    states[1] = 255; states[1, oh // 3: oh // 3 + 2, ow // 4: ow // 4 + 3] = 254; states[1, -1, -1] = 0      # almost uniform: one tile must NOT take the shortcut
    states[2] = rng.integers(0, 2, (oh, ow)).astype(np.uint8) * 255; states[2, :, : ow // 2] = 0             # half noise, half uniform 0
    states[3] = rng.integers(0, 256, (oh, ow), dtype=np.uint8)                                             # no uniform tile at all
    states[3, oh // 2:, ow // 2:] = 1
    for mg in (mg_u, mg_g):
        mg.ofinal().copy_(_dev(states))
        mg.run_stage(3, n=n)                                       # mask_tile_k<false>
    mu, mgm = mg_u.masks().cpu().numpy(), mg_g.masks().cpu().numpy()
    assert np.array_equal(mu, mgm)
    rx, ry, rw, rh = i["roi"]
    qx, qy, qw, qh = i["in_roi"]
    for k in range(n):
        want = np.full((H, W), 255, np.uint8)
        want[ry:ry + rh, rx:rx + rw] = oracle.blur5(oracle.resize_linear(np.ascontiguousarray(states[k, qy:qy + qh, qx:qx + qw]), rw, rh))
        assert np.array_equal(mu[k], want), "state %d" % k
    # the whole step, frame by frame: identical temporal state on both contexts → identical bytes out of the fused kernel's two paths
    mg_u.reset(); mg_g.reset()
    bg = _dev(synth.random_u8((n, H, W, 3), 81))
    out_u = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    out_g = torch.empty_like(out_u)
    y_u = torch.empty((n, H, W, 2), dtype=torch.uint8, device="cuda")
    y_g = torch.empty_like(y_u)
    frames = _dev(np.stack([synth.frame(W, H, s, 0) for s in range(n)]))
    for t in range(5):
        kw = [dict(), dict(flip_h=True), dict(yuyv=True), dict(flip_v=True, no_mask=True), dict()][t]
        if W % 2 and kw.get("yuyv"):
            kw = dict()
        a, b = (y_u, y_g) if kw.get("yuyv") else (out_u, out_g)
        mg_u.step_ex(frames, bg, a, **kw)
        mg_g.step_ex(frames, bg, b, **kw)
        assert torch.equal(a, b), "frame %d %s: %d bytes differ" % (t, kw, int((a != b).sum()))
        assert torch.equal(mg_u.masks(), mg_g.masks()) and torch.equal(mg_u.ofinal(), mg_g.ofinal())
    of = mg_u.ofinal().cpu().numpy()
    assert set(np.unique(of).tolist()) <= {0, 255}                  # steady state reached: the last frames DID run the shortcut
    mg_u.close(); mg_g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("key,res", [("lite", VGA), ("mlkit", VGA)])
def test_step_in_place_like_the_reference(bs, oracle, key, res):
    """The reference flips and composites `raw` IN PLACE (app/deepseg.cc:661-673): d_out == d_frames.  Round 3's fused kernel stored the flipped composite to
    addresses another tile had not read yet (advisor, medium).  Every flag combination must give the bytes of the out-of-place call; partial overlap is refused."""
    import torch
    from backscrub_amd import synth
    W, H = res
    n = 3
    mg_a = bs.MaskGen(model_path(key), W, H, n_streams=n)
    mg_b = bs.MaskGen(model_path(key), W, H, n_streams=n)
    bg = _dev(synth.random_u8((n, H, W, 3), 79))
    want = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    t = 0
    for fh, fv in ((False, False), (True, False), (False, True), (True, True)):
        frames = _dev(np.stack([synth.frame(W, H, i, t) for i in range(n)]))
        t += 1
        inplace = frames.clone()
        mg_a.step_ex(frames, bg, want, flip_h=fh, flip_v=fv)
        mg_b.step_ex(inplace, bg, inplace, flip_h=fh, flip_v=fv)
        assert torch.equal(inplace, want), "in place, flip_h=%s flip_v=%s: %d bytes differ" % (fh, fv, int((inplace != want).sum()))
        assert torch.equal(mg_a.masks(), mg_b.masks())
    # YUYV in place: 2 B/px written over the 3 B/px frames
    frames = _dev(np.stack([synth.frame(W, H, i, t) for i in range(n)]))
    buf = frames.clone()
    want2 = torch.empty((n, H, W, 2), dtype=torch.uint8, device="cuda")
    mg_a.step_ex(frames, bg, want2, yuyv=True, flip_h=True)
    out_view = buf.view(-1)[: n * H * W * 2].view(n, H, W, 2)
    mg_b.step_ex(buf, bg, out_view, yuyv=True, flip_h=True)
    assert torch.equal(out_view, want2)
    # partial overlap: refused
    big = torch.empty((n + 1, H, W, 3), dtype=torch.uint8, device="cuda")
    shifted = big.view(-1)[W * 3 * 8: W * 3 * 8 + n * H * W * 3].view(n, H, W, 3)
    with pytest.raises(bs.BsxError):
        mg_b.step_ex(big[:n], bg, shifted)
    mg_a.close()
    mg_b.close()


@pytest.mark.parametrize("key,res", [("lite", VGA), ("mlkit", VGA), ("full", (1280, 720)), ("lite", (322, 242))])
def test_step_with_own_blur_as_background(bs, oracle, key, res):
    """BSX_STEP_BGBLUR(ksize): `-p bgblur:<n>` without `-b` (app/deepseg.cc:652-661) in one pass over the frames — the blurred tile is composited out of LDS.
    Bit-identical to bsx_gaussian_blur_bgr into a per-stream background + bsx_step_batch, on the four flip-test geometries (the last one, width % 4 != 0,
    takes the two-pass form inside the library), for three kernel sizes, with flags that force the two-pass form, and against the oracle's blur + blend."""
    import torch
    from backscrub_amd import synth
    W, H = res
    n = 3
    mg_a = bs.MaskGen(model_path(key), W, H, n_streams=n)
    mg_b = bs.MaskGen(model_path(key), W, H, n_streams=n)
    out_a = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    out_b = torch.empty_like(out_a)
    for t, ksize in enumerate((25, 3, 31, 25)):
        frames = _dev(np.stack([synth.frame(W, H, i, t) for i in range(n)]))
        mg_a.step(frames, mg_a.gaussian_blur(frames, ksize), out_a)
        mg_b.step_ex(frames, None, out_b, bgblur=ksize)
        assert torch.equal(out_a, out_b), "t=%d ksize=%d: %d bytes differ" % (t, ksize, int((out_a != out_b).sum()))
        assert torch.equal(mg_a.masks(), mg_b.masks())
    f0 = frames[0].cpu().numpy()
    assert np.array_equal(out_b[0].cpu().numpy(), oracle.alpha_blend(oracle.gaussian_blur(f0, 25), f0, mg_b.masks()[0].cpu().numpy()))
    # flags that the single pass does not serve: same bytes through the library's own two-pass form
    mg_a.step_ex(frames, mg_a.gaussian_blur(frames, 7), out_a, flip_h=True)
    mg_b.step_ex(frames, None, out_b, flip_h=True, bgblur=7)
    assert torch.equal(out_a, out_b)
    mg_a.step(frames, frames, out_a)                                     # ksize 1: the blur is the frame itself
    mg_b.step_ex(frames, None, out_b, bgblur=1)
    assert torch.equal(out_a, out_b) and torch.equal(out_b, frames)
    for bad in (4, 33):
        with pytest.raises(bs.BsxError):
            mg_b.step_ex(frames, None, out_b, bgblur=bad)
    with pytest.raises(bs.BsxError):
        mg_b.step_ex(frames, None, frames, bgblur=5)                     # in place: the blur reads neighbours the composite would overwrite
    mg_a.close()
    mg_b.close()


@pytest.mark.parametrize("key,res", [("lite", VGA), ("mlkit", VGA)])
def test_step_without_storing_the_mask(bs, key, res):
    """BSX_STEP_NO_MASK: the same composite bytes as bsx_step_batch, the temporal state advances identically (the next frames agree too), and the
    full-resolution mask buffer is left as the last storing call wrote it."""
    from backscrub_amd import synth
    W, H = res
    n = 3
    mg_a = bs.MaskGen(model_path(key), W, H, n_streams=n)
    mg_b = bs.MaskGen(model_path(key), W, H, n_streams=n)
    bg = _dev(synth.random_u8((H, W, 3), 79))
    out_a = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    out_b = torch.empty_like(out_a)
    before = mg_b.masks().clone()
    for t in range(4):
        frames = _dev(np.stack([synth.frame(W, H, i, t) for i in range(n)]))
        mg_a.step(frames, bg, out_a)
        mg_b.step_ex(frames, bg, out_b, no_mask=(t != 3))
        assert torch.equal(out_a, out_b), "t=%d" % t
        if t != 3:
            assert torch.equal(mg_b.masks(), before)             # untouched: all 255, as bsx_new left it
    assert torch.equal(mg_a.masks(), mg_b.masks())                # the storing call writes the same mask the other context has
    mg_a.close()
    mg_b.close()


def test_yuyv_to_bgr_matches_oracle(bs, oracle):
    from backscrub_amd import synth
    img = synth.random_u8((2, 480, 640, 2), 13)
    mg = bs.MaskGen(synthetic_model_path("lite"), 640, 480, n_streams=1)
    got = mg.yuyv_to_bgr(_dev(img)).cpu().numpy()
    for i in range(2):
        assert np.array_equal(got[i], oracle.yuyv_to_bgr(img[i]))
    mg.close()


# --------------------------------------------------------------------------------------------
# size-independent properties at the BASELINE batch size (256 VGA streams, segm_lite)
# --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("ksize,res", [(25, VGA), (3, (322, 242)), (5, (64, 16)), (7, (65, 17)), (9, (33, 47)), (31, (130, 70)), (1, (40, 30)), (25, (20, 12))])
def test_gaussian_blur_matches_oracle(bs, oracle, ksize, res):
    """bsx_gaussian_blur_bgr (the -p bgblur:<n> step, app/deepseg.cc:657-658) is integer arithmetic: bit-exact against the oracle,
    including images smaller than the kernel radius (multiple reflections) and sizes that are not multiples of the 64x16 tile."""
    from backscrub_amd import synth
    W, H = res
    img = synth.random_u8((2, H, W, 3), 60 + ksize)
    img[0, : H // 3] = 255
    mg = bs.MaskGen(synthetic_model_path("lite"), 640, 480, n_streams=1)
    got = mg.gaussian_blur(_dev(img), ksize).cpu().numpy()
    for i in range(2):
        want = oracle.gaussian_blur(img[i], ksize)
        assert np.array_equal(got[i], want), "ksize %d image %d: %d bytes differ" % (ksize, i, (got[i] != want).sum())
    with pytest.raises(bs.BsxError):
        mg.gaussian_blur(_dev(img), 4)
    with pytest.raises(bs.BsxError):
        mg.gaussian_blur(_dev(img), 33)
    mg.close()


def test_blur_own_frame_as_background(bs, oracle):
    """The reference's most used mode (-p bgblur without -b, app/deepseg.cc:652-661): background = GaussianBlur(camera frame),
    then alpha_blend — as bsx_gaussian_blur_bgr + bsx_step_batch with one background per stream."""
    from backscrub_amd import synth
    path = model_path("lite")
    W, H = VGA
    n = 2
    mg = bs.MaskGen(path, W, H, n_streams=n)
    oc = [oracle.Ctx(path, W, H) for _ in range(n)]
    out = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    for t in range(3):
        frames = np.stack([synth.frame(W, H, s, t) for s in range(n)])
        d_frames = _dev(frames)
        d_bg = mg.gaussian_blur(d_frames, 25)
        mg.step(d_frames, d_bg, out)
        got_m, got_o = mg.masks().cpu().numpy(), out.cpu().numpy()
        for i in range(n):
            want_m = oc[i].process(frames[i])
            assert _iou_fg(got_m[i], want_m) >= 0.999
            if np.array_equal(got_m[i], want_m):
                assert np.array_equal(got_o[i], oracle.alpha_blend(oracle.gaussian_blur(frames[i], 25), frames[i], want_m))
    for c in oc:
        c.close()
    mg.close()


@pytest.mark.parametrize("code", [0, 1, -1])
def test_flip_matches_oracle(bs, oracle, code):
    from backscrub_amd import synth
    mg = bs.MaskGen(synthetic_model_path("lite"), 64, 48, n_streams=1)
    img = synth.random_u8((3, 37, 53, 3), 31)
    got = mg.flip_bgr(_dev(img), code).cpu().numpy()
    for i in range(3):
        assert np.array_equal(got[i], oracle.flip_bgr(img[i], code))
    mg.close()


def test_full_batch_properties(bs, oracle):
    from backscrub_amd import synth
    W, H = VGA
    n = 256
    path = model_path("lite")
    mg = bs.MaskGen(path, W, H, n_streams=n)
    base = synth.frames(8, W, H)
    frames = np.concatenate([base] * (n // 8))           # stream i carries scene i % 8
    d_frames = _dev(frames)
    bg = synth.background(W, H)
    d_bg = _dev(bg)
    out = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
    for _ in range(4):
        mg.step(d_frames, d_bg, out)
    masks = mg.masks().cpu().numpy()
    # streams are independent and deterministic: identical inputs → identical state, wherever they sit in the batch
    for i in range(8, n):
        assert np.array_equal(masks[i], masks[i % 8]), "stream %d differs from its twin" % i
    # IIR steady state after >= 3 identical frames: model-resolution mask is exactly 0x00 / 0xFF
    of = mg.ofinal().cpu().numpy()
    assert set(np.unique(of).tolist()) <= {0, 255}
    # idempotence at steady state
    before = masks.copy()
    mg.step(d_frames, d_bg, out)
    assert np.array_equal(mg.masks().cpu().numpy(), before)
    # blend endpoints: mask 255 → background, mask 0 → camera frame
    o = out.cpu().numpy()
    m = mg.masks().cpu().numpy()
    assert np.array_equal(o[m == 255], np.broadcast_to(bg, o.shape)[m == 255])
    assert np.array_equal(o[m == 0], frames[m == 0])
    # and the oracle agrees on the 8 distinct scenes
    for i in range(8):
        oc = oracle.Ctx(path, W, H)
        for _ in range(5):
            want = oc.process(frames[i])
        assert _iou_fg(m[i], want) >= 0.999
        oc.close()
    mg.close()


# --------------------------------------------------------------------------------------------
# ragged / odd inputs and argument errors
# --------------------------------------------------------------------------------------------
def test_partial_batches_and_odd_frame_sizes(bs, oracle):
    """n < n_streams batches touch only their streams; odd frame sizes (unaligned rows, partial tiles) stay exact."""
    from backscrub_amd import synth
    path = model_path("lite")
    for (W, H) in ((322, 242), (641, 479), (160, 96)):
        cap = 5
        mg = bs.MaskGen(path, W, H, n_streams=cap)
        oc = [oracle.Ctx(path, W, H) for _ in range(cap)]
        bg = synth.background(W, H)
        out = torch.empty((cap, H, W, 3), dtype=torch.uint8, device="cuda")
        for t, n in enumerate((5, 2, 3)):                       # ragged: later batches use only the first n streams
            frames = np.stack([synth.frame(W, H, s, t) for s in range(n)])
            mg.step(_dev(frames), _dev(bg), out[:n])
            got_m = mg.masks().cpu().numpy()
            got_o = out.cpu().numpy()
            for i in range(n):
                want = oc[i].process(frames[i])
                assert _iou_fg(got_m[i], want) >= 0.999, (W, H, t, i)
                if np.array_equal(got_m[i], want):
                    assert np.array_equal(got_o[i], oracle.alpha_blend(bg, frames[i], want))
            for i in range(n, cap):                             # untouched streams keep their previous state exactly
                assert np.array_equal(got_m[i], oc[i].mask())
        for c in oc:
            c.close()
        mg.close()


def test_argument_errors(bs):
    from backscrub_amd import synth
    W, H = VGA
    mg = bs.MaskGen(synthetic_model_path("lite"), W, H, n_streams=2)
    f3 = _dev(synth.frames(3, W, H, distinct=1))
    with pytest.raises(bs.BsxError):
        mg.process_batch(f3)                                    # batch larger than n_streams
    with pytest.raises(bs.BsxError):
        mg.process_batch(_dev(synth.frames(1, W // 2, H, distinct=1)))   # wrong geometry
    with pytest.raises(bs.BsxError):
        mg.process_host(np.zeros((H, W // 2, 3), np.uint8))
    assert bs.lib().bsx_process_batch(None, None, 1, None, None) == -1      # BSX_EINVAL on a NULL context (libbackscrub.cc:280)
    assert bs.lib().bsx_process_batch(mg.h, None, 1, None, None) == -1
    mg.close()


def test_release_library_ignores_the_work_skipping_switches(bs, monkeypatch):
    """BSX_SEG_SKIP makes the segment kernels leave phases out (a timing experiment: the logits are then WRONG).  The release library must not even read it:
    same logits, bit for bit, with and without the variable — while the debug build (libbsx_dbg.so, test infrastructure) honours it."""
    from backscrub_amd import api, build, synth
    path = model_path("lite")
    W, H = VGA
    f = _dev(np.stack([synth.frame(W, H, 5), synth.frame(W, H, 6)]))

    def logits():
        mg = bs.MaskGen(path, W, H, n_streams=2)
        plan = mg.plan()
        mg.run_stage(0, f)
        mg.run_stage(1, n=2)
        out = mg.output_tensor().clone()
        mg.close()
        return out, plan
    for k in ("BSX_SEG_SKIP", "BSX_SEG_GATE_SKIP", "BSX_PROGRAM_NOP"):
        monkeypatch.delenv(k, raising=False)
    assert os.path.samefile(api.lib_path(), build.LIB)
    want, plan = logits()
    assert "program execution: specialised kernel" in plan
    monkeypatch.setenv("BSX_SEG_SKIP", "255,255,255,255")
    monkeypatch.setenv("BSX_SEG_GATE_SKIP", "1")
    monkeypatch.setenv("BSX_PROGRAM_NOP", "1")
    got, plan2 = logits()
    assert torch.equal(got, want) and plan2 == plan, "the release library read a debug switch"
    saved = api._LIB
    try:                                          # the same variables under the debug build: the experiment takes effect — every micro-op of the middle program is a
        api._LIB = None                           # no-op (kind 99), which the generator refuses, so the plan falls back to the interpreter.  (The logits are not
        monkeypatch.setenv("BSX_LIBRARY", build.LIB_DBG)      # compared: kernels that skip their stores leave whatever the allocation held — usually the run before.)
        _, plan3 = logits()
        assert "micro-op kind 99" in plan3 and "program execution: specialised kernel" not in plan3
    finally:
        monkeypatch.delenv("BSX_LIBRARY", raising=False)
        api._LIB = saved


@pytest.mark.parametrize("key,res", [("lite", VGA), ("mlkit", HD), ("full", HD)])
def test_specialised_segment_kernels_equal_the_ahead_of_time_ones(bs, key, res, monkeypatch, debug_switches):
    """The product path runs the segment kernels hipRTC compiled for the loaded graph (csrc/gen_seg.cpp); the ahead-of-time kernels of kernels_seg.hip — same source, the
    descriptor a kernel argument — are the fallback.  Same arithmetic in the same order: temporal state, masks and composites identical bit for bit over three steps
    (BSX_NO_SEG_RTC=1, debug build, selects the ahead-of-time kernels), and the plan says which ones ran."""
    from backscrub_amd import synth
    W, H = res
    n = 3
    bg = _dev(synth.background(W, H))
    got = []
    for aot in (False, True):
        if aot:
            monkeypatch.setenv("BSX_NO_SEG_RTC", "1")
        else:
            monkeypatch.delenv("BSX_NO_SEG_RTC", raising=False)
        mg = bs.MaskGen(model_path(key), W, H, n_streams=n)
        assert ("segment execution: specialised kernels (hipRTC" in mg.plan()) == (not aot), mg.plan()[-400:]
        out = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
        snaps = []
        for t in range(3):
            frames = np.stack([synth.frame(W, H, i, t) for i in range(n)])
            frames[n - 1] = synth.random_u8((H, W, 3), 50 + t)
            mg.step(_dev(frames), bg, out)
            snaps.append((mg.ofinal().clone(), mg.masks().clone(), out.clone()))
        got.append(snaps)
        mg.close()
    monkeypatch.delenv("BSX_NO_SEG_RTC", raising=False)
    for t in range(3):
        for a, b in zip(got[0][t], got[1][t]):
            assert torch.equal(a, b), "step %d" % t
