"""Pins for the CPU oracle's integer / image stages: independent numpy re-derivations of the published
OpenCV 8-bit formulas, exhaustive integer identities, the geometry table of SURVEY.md §8, and the
edge cases the domain has (tiny / ragged sizes, identity resize, exact 2x area case, IIR steady state)."""
import numpy as np
import pytest

from conftest import MODEL_KEYS, model_path


# ---- geometry (lib/libbackscrub.cc:234-246) ------------------------------------------------------
GEOMETRY = [  # frame, model key, roidim, in_roidim  — SURVEY.md §8 table
    ((640, 480), "mlkit", (80, 0, 480, 480), (0, 0, 256, 256)),
    ((640, 480), "lite", (0, 0, 640, 480), (16, 0, 128, 96)),
    ((640, 480), "deeplab", (80, 0, 480, 480), (0, 0, 257, 257)),
    ((1280, 720), "mlkit", (280, 0, 720, 720), (0, 0, 256, 256)),
    ((1280, 720), "full", (0, 0, 1280, 720), (0, 0, 256, 144)),
    ((1280, 720), "lite", (40, 0, 1200, 720), (0, 0, 160, 96)),
    ((640, 480), "full", (0, 0, 640, 480), (32, 0, 192, 144)),
]


@pytest.mark.parametrize("res,key,roi,in_roi", GEOMETRY)
def test_geometry_table(oracle, res, key, roi, in_roi):
    c = oracle.Ctx(model_path(key), *res)
    assert c.roidim == roi and c.in_roidim == in_roi
    assert c.mask().min() == 255          # :248 mask starts all-background
    c.close()


# ---- resize: independent vectorised numpy statement of OpenCV's 8u INTER_LINEAR -------------------
def _np_resize(src, dw, dh):
    sh, sw = src.shape[:2]
    if (sw, sh) == (dw, dh):
        return src.copy()
    s = src.astype(np.int64).reshape(sh, sw, -1)
    sx_, sy_ = 1.0 / (dw / sw), 1.0 / (dh / sh)
    if sx_ == 2.0 and sy_ == 2.0:
        out = (s[0::2, 0::2] + s[0::2, 1::2] + s[1::2, 0::2] + s[1::2, 1::2] + 2) >> 2
        return out.astype(np.uint8).reshape((dh, dw) + src.shape[2:])

    def coeffs(d, sc, n, clamp):
        f = ((np.arange(d) + 0.5) * sc - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int64)
        f = (f - i.astype(np.float32)).astype(np.float32)
        if clamp:
            lo, hi = i < 0, i >= n - 1
            f[lo | hi] = 0
            i[lo] = 0
            i[hi] = n - 1
        a1 = np.rint(f * np.float32(2048)).astype(np.int64)
        a0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        return i, a0, a1

    xi, xa0, xa1 = coeffs(dw, sx_, sw, True)
    yi, yb0, yb1 = coeffs(dh, sy_, sh, False)
    xi1 = np.minimum(xi + 1, sw - 1)
    y0, y1 = np.clip(yi, 0, sh - 1), np.clip(yi + 1, 0, sh - 1)
    h = s[:, xi] * xa0[None, :, None] + s[:, xi1] * xa1[None, :, None]
    r0, r1 = h[y0], h[y1]
    out = (((yb0[:, None, None] * (r0 >> 4)) >> 16) + ((yb1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8).reshape((dh, dw) + src.shape[2:])


@pytest.mark.parametrize("src,dst,cn", [((480, 480), (256, 256), 3), ((640, 480), (128, 96), 3), ((128, 96), (640, 480), 1),
                                        ((256, 144), (1280, 720), 1), ((1200, 859), (640, 480), 3), ((1280, 960), (640, 480), 3),
                                        ((7, 5), (31, 17), 1), ((31, 17), (7, 5), 3), ((2, 2), (9, 9), 1), ((64, 48), (64, 48), 3)])
def test_resize_matches_numpy_restatement(oracle, src, dst, cn):
    rng = np.random.default_rng(src[0] * 131 + dst[0])
    img = rng.integers(0, 256, (src[1], src[0], cn) if cn > 1 else (src[1], src[0]), dtype=np.uint8)
    got = oracle.resize_linear(img, dst[0], dst[1])
    assert np.array_equal(got, _np_resize(img, dst[0], dst[1]))


def test_resize_constant_image_stays_constant(oracle):
    for v in (0, 1, 127, 254, 255):
        img = np.full((37, 53, 3), v, np.uint8)
        assert (oracle.resize_linear(img, 200, 150) == v).all()
        assert (oracle.resize_linear(img, 11, 9) == v).all()


# ---- blur 5x5: (s+12)/25 with REFLECT_101 ------------------------------------------------------------
def test_blur_is_rounded_box_mean(oracle):
    rng = np.random.default_rng(5)
    for shape in ((480, 640), (5, 7), (9, 3), (64, 64)):
        img = rng.integers(0, 256, shape, dtype=np.uint8)
        p = np.pad(img.astype(np.int64), 2, mode="reflect")
        s = sum(p[j:j + shape[0], i:i + shape[1]] for j in range(5) for i in range(5))
        assert np.array_equal(oracle.blur5(img), ((s + 12) // 25).astype(np.uint8))
    # the identity round(s/25) == (s+12)//25 over the whole range (no ties: 25 is odd)
    s = np.arange(0, 25 * 255 + 1)
    assert np.array_equal(np.rint(s * (1.0 / 25)).astype(np.int64), (s + 12) // 25)


# ---- alpha blend (app/deepseg.cc:108-134) --------------------------------------------------------------
def test_blend_truncating_divide_and_endpoints(oracle):
    rng = np.random.default_rng(1)
    bg, fr = rng.integers(0, 256, (2, 33, 47, 3), dtype=np.uint8)
    m = rng.integers(0, 256, (33, 47), dtype=np.uint8)
    m[0] = 255
    m[1] = 0
    out = oracle.alpha_blend(bg, fr, m)
    want = (bg.astype(np.int64) * m[..., None] + fr.astype(np.int64) * (255 - m[..., None].astype(np.int64))) // 255
    assert np.array_equal(out, want.astype(np.uint8))
    assert np.array_equal(out[0], bg[0]) and np.array_equal(out[1], fr[1])   # 255 ⇒ background, 0 ⇒ camera


def test_packed_blend_identities_hold_on_the_whole_range():
    """The integer forms the HIP blend uses (csrc/kernels_img.hip: pk_blend), over every value they can meet: t = a*m + b*(255-m) <= 65025 fits a u16 lane;
    round 4's floor(t / 255) == (t + 1 + (t >> 8)) >> 8 and round 5's — with u = t + 1 formed by the multiply-adds themselves — floor((u - 1) / 255) == (u + (u >> 8)) >> 8,
    every intermediate below 2^16; and every (a, b, m) byte triple through the u-form equals deepseg.cc:108-134's truncating divide."""
    t = np.arange(0, 255 * 255 + 1, dtype=np.int64)
    assert np.array_equal((t + 1 + (t >> 8)) >> 8, t // 255) and int((t + 1 + (t >> 8)).max()) < 65536
    u = t + 1
    assert np.array_equal((u + (u >> 8)) >> 8, t // 255) and int((u + (u >> 8)).max()) == 65280 and int(u.max()) == 65026
    a = np.arange(256, dtype=np.int64)[:, None, None]
    b = np.arange(256, dtype=np.int64)[None, :, None]
    m = np.arange(256, dtype=np.int64)[None, None, :]
    uu = a * m + 1                                   # first v_pk_mad_u16
    assert int(uu.max()) <= 65026
    uu = b * (255 - m) + uu                          # second
    assert int(uu.max()) <= 65026 and int((255 ^ m).max()) == 255 and np.array_equal(255 ^ m, 255 - m)      # 0x00ff00ff ^ m == 255 - m per half
    assert np.array_equal((uu + (uu >> 8)) >> 8, (a * m + b * (255 - m)) // 255)


# ---- decode + IIR (lib/libbackscrub.cc:317-357) -----------------------------------------------------------
def test_iir_reaches_steady_state_in_three_frames(oracle):
    prob = np.full((4, 4, 1), 0.1, np.float32)          # "not a person" → val 255
    o = np.zeros((4, 4), np.uint8)
    seq = []
    for _ in range(4):
        o = oracle.decode_iir(2, prob, o)
        seq.append(int(o[0, 0]))
    assert seq == [0xE0, 0xFC, 0xFF, 0xFF]
    prob[:] = 0.9                                        # person → val 0
    seq = []
    for _ in range(4):
        o = oracle.decode_iir(2, prob, o)
        seq.append(int(o[0, 0]))
    assert seq == [0x1F, 0x03, 0x00, 0x00]


def test_decode_rules(oracle):
    z = np.zeros((1, 6), np.uint8)
    # MLKit: strictly greater than the DOUBLE literal 0.65 (float32(0.65) < 0.65 → background)
    p = np.array([[0.65, np.nextafter(np.float32(0.65), np.float32(1)), 0.6499, 0.66, np.nan, 1.0]], np.float32)[..., None]
    assert (oracle.decode_iir(2, p, z)[0] >> 5).tolist() == [7, 0, 7, 0, 7, 0]
    # Meet: softmax-2 compare; ties, overflow (inf/inf = NaN → background) and NaN go to 255
    l = np.array([[[0, 1], [1, 0], [2, 2], [100, 101], [-200, -201], [np.nan, 1]]], np.float32)
    assert (oracle.decode_iir(3, l, z)[0] >> 5).tolist() == [0, 7, 7, 7, 7, 7]
    # DeepLab: first maximum wins, person = class 15, everything below -10000 selects class 0
    d = np.zeros((1, 3, 21), np.float32)
    d[0, 0, 15] = 1
    d[0, 1, 15] = 1
    d[0, 1, 3] = 1          # earlier class with the same value wins
    d[0, 2, :] = -20000      # nothing beats the initial -10000 → maxpos stays 0
    assert (oracle.decode_iir(1, d, np.zeros((1, 3), np.uint8))[0] >> 5).tolist() == [0, 7, 7]


# ---- bilateral: independent float32 numpy statement in the same tap order -------------------------------------
def test_bilateral_matches_numpy_restatement(oracle):
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (23, 31, 3), dtype=np.uint8)
    img[:8] = (rng.integers(0, 256, (8, 31, 1)) // 8 * 8).astype(np.uint8)   # smooth-ish part
    got = oracle.bilateral(img)
    lut = np.exp(np.arange(768, dtype=np.float64) ** 2 * (-0.5 / 100.0 ** 2)).astype(np.float32)
    taps = [(i, j) for i in range(-2, 3) for j in range(-2, 3) if np.sqrt(i * i + j * j) <= 2]
    assert len(taps) == 13
    p = np.pad(img, ((2, 2), (2, 2), (0, 0)), mode="reflect").astype(np.int32)
    c = img.astype(np.int32)
    acc = np.zeros(img.shape, np.float32)
    ws = np.zeros(img.shape[:2], np.float32)
    for i, j in taps:
        nb = p[2 + i:2 + i + img.shape[0], 2 + j:2 + j + img.shape[1]]
        w = np.float32(np.exp((i * i + j * j) * (-0.5 / 100.0 ** 2))) * lut[np.abs(nb - c).sum(-1)]
        acc = acc + nb.astype(np.float32) * w[..., None]
        ws = ws + w
    want = np.rint(acc * (np.float32(1) / ws)[..., None]).astype(np.uint8)
    assert np.array_equal(got, want)
    flat = np.full((9, 9, 3), 77, np.uint8)
    assert np.array_equal(oracle.bilateral(flat), flat)


# ---- YUYV packer (app/deepseg.cc:87-106) --------------------------------------------------------------------------
def test_yuyv_layout_and_formula(oracle):
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (4, 8, 3), dtype=np.uint8)
    out = oracle.bgr_to_yuyv(img).reshape(-1, 4)
    a = img.reshape(-1, 3).astype(np.int64)
    R, G, B = a[:, 0], a[:, 1], a[:, 2]                    # RGB2YUV applied to BGR-ordered bytes
    Y = (R * 4899 + G * 9617 + B * 1868 + 8192) >> 14
    U = np.clip(((B - Y) * 8061 + (128 << 14) + 8192) >> 14, 0, 255)
    V = np.clip(((R - Y) * 14369 + (128 << 14) + 8192) >> 14, 0, 255)
    assert np.array_equal(out[:, 0], Y[0::2]) and np.array_equal(out[:, 2], Y[1::2])
    assert np.array_equal(out[:, 1], (V[0::2] + V[1::2]) // 2) and np.array_equal(out[:, 3], (U[0::2] + U[1::2]) // 2)


# ---- whole context: persistent border, determinism ------------------------------------------------------------------
def test_context_is_deterministic_and_border_persists(oracle):
    from backscrub_amd import synth
    path = model_path("mlkit")
    a, b = oracle.Ctx(path, 640, 480), oracle.Ctx(path, 640, 480)
    for t in range(3):
        f = synth.frame(640, 480, 1, t)
        ma, mb = a.process(f), b.process(f)
        assert np.array_equal(ma, mb)
    x, _, w, _ = a.roidim
    assert (ma[:, :x] == 255).all() and (ma[:, x + w:] == 255).all()
    a.close()
    b.close()


def test_yuyv_to_bgr_formula_and_roundtrip_bound(oracle):
    rng = np.random.default_rng(4)
    yuyv = rng.integers(0, 256, (6, 10, 2), dtype=np.uint8)
    got = oracle.yuyv_to_bgr(yuyv).reshape(-1, 2, 3).astype(np.int64)
    p = yuyv.reshape(-1, 4).astype(np.int64)
    u, v = p[:, 1] - 128, p[:, 3] - 128
    for k, yy in enumerate((p[:, 0], p[:, 2])):
        y = np.maximum(0, yy - 16) * 1220542
        b = np.clip((y + (1 << 19) + 2116026 * u) >> 20, 0, 255)
        g = np.clip((y + (1 << 19) - 852492 * v - 409993 * u) >> 20, 0, 255)
        r = np.clip((y + (1 << 19) + 1673527 * v) >> 20, 0, 255)
        assert np.array_equal(got[:, k, 0], b) and np.array_equal(got[:, k, 1], g) and np.array_equal(got[:, k, 2], r)
    # grey ramp: U = V = 128 → B = G = R = clamp(1.164 * (Y - 16))
    grey = np.stack([np.arange(256, dtype=np.uint8), np.full(256, 128, np.uint8)], -1).reshape(1, 256, 2)
    out = oracle.yuyv_to_bgr(grey)[0]
    assert (out[:, 0] == out[:, 1]).all() and (out[:, 1] == out[:, 2]).all()
    assert out[16, 0] == 0 and out[235, 0] == 255


# ---- cv::GaussianBlur 8-bit fixed-point path (app/deepseg.cc:657-658, -p bgblur:<n>) -------------------------------------------
def _gauss_numpy(img, n):
    """Independent numpy statement of OpenCV's 8-bit Gaussian: ufixedpoint16 coefficients, u16 horizontal / u32 vertical passes."""
    if n == 1:
        c = np.array([256])
    elif n == 3:
        c = np.array([64, 128, 64])
    elif n == 5:
        c = np.array([16, 64, 96, 64, 16])
    elif n == 7:
        c = np.array([8, 28, 56, 72, 56, 28, 8])
    else:
        sigma = ((n - 1) * 0.5 - 1) * 0.3 + 0.8
        x = np.arange(1 - n, n, 2, dtype=np.float64)
        v = np.exp(x * x * (-0.125 / (sigma * sigma)))
        c = np.rint(v / v.sum() * 256.0).astype(np.int64)
    r = n // 2
    pad = np.pad(img.astype(np.int64), ((r, r), (r, r), (0, 0)), mode="reflect")      # numpy "reflect" == BORDER_REFLECT_101
    h = sum(c[k] * pad[:, k:k + img.shape[1]] for k in range(n))
    h = np.minimum(h, 0xFFFF)
    v = sum(c[k] * h[k:k + img.shape[0]] for k in range(n))
    v = np.minimum(v, 0xFFFFFFFF)
    return np.minimum((v + (1 << 15)) >> 16, 255).astype(np.uint8)


@pytest.mark.parametrize("n", [1, 3, 5, 7, 9, 25, 31])
def test_gaussian_blur_restatement(oracle, n):
    from backscrub_amd import synth
    img = synth.random_u8((37, 53, 3), 40 + n)
    img[:max(6, n), :max(6, n)] = 255                   # saturation corner: with sum(c) = 257 the result must clamp at 255, not wrap
    got = oracle.gaussian_blur(img, n)
    assert np.array_equal(got, _gauss_numpy(img, n))
    c = oracle.gaussian_coeffs(n)
    assert 255 <= int(c.sum()) <= 257 and (c == c[::-1]).all() and (n == 1 or int(c.max()) < 256)
    if n >= 3:
        # it IS a Gaussian of the sigma OpenCV derives from the kernel size: within 3 grey levels of the float filter
        from scipy.ndimage import gaussian_filter1d
        sigma = 0.3 * ((n - 1) * 0.5 - 1) + 0.8
        f = img.astype(np.float64)
        for ax in (0, 1):
            f = gaussian_filter1d(f, sigma, axis=ax, mode="mirror", truncate=(n // 2) / sigma)
        if n > 7:
            assert np.abs(f - got).max() <= 3.0     # 8-bit coefficients (sum 256 or 257) on a pure-noise image
    assert got[0, 0].min() == 255


@pytest.mark.parametrize("n", list(range(3, 32, 2)))
def test_gauss_kernel_coefficient_words_are_the_oracle_taps_delayed(oracle, n):
    """gauss_blur_k never realigns its data: the 4 neighbouring outputs of a horizontal item (2 of a vertical item) multiply the SAME aligned words with the tap sequence
    delayed by 0..3 bytes (0..1 halves), plus `shift` more bytes when the LDS planes start left of the tile (4-pixel staging).  The words the launcher hands to the
    kernel (bsx_debug_gauss_coeffs, host only) must therefore be exactly the oracle's taps at those delays and zero everywhere else — for every kernel size and shift,
    and inside the word counts the kernel templates read (NT = ceil((n + 3 + shift) / 4) bytes-words, 2 NT - 1 half-words)."""
    import backscrub_amd.api as api
    taps = oracle.gaussian_coeffs(n).astype(np.int64)
    assert 240 <= taps.sum() <= 257 and (taps <= 255).all()              # per-tap rounding: the sum drifts from 256 (252 at ksize 29); the kernel needs <= 257
    for shift in range(4):
        if n + shift > 32:
            with pytest.raises(api.BsxError):
                api.gauss_coeff_words(n, shift)
            continue
        c4, c2 = api.gauss_coeff_words(n, shift)
        nt = (n + 3 + shift + 3) // 4
        assert 2 <= nt <= 9
        for j in range(4):
            b = c4[j].view(np.uint8).astype(np.int64)                     # little endian: byte k of the sequence = tap k - (j + shift)
            want = np.zeros(36, np.int64)
            want[j + shift: j + shift + n] = taps
            assert np.array_equal(b, want), "ksize %d shift %d phase %d" % (n, shift, j)
            assert not b[4 * nt:].any()                                    # nothing beyond the words the horizontal pass reads
        for h in range(2):
            w = c2[h].view(np.uint16).astype(np.int64)
            want = np.zeros(34, np.int64)
            want[h: h + n] = taps
            assert np.array_equal(w, want), "ksize %d half-phase %d" % (n, h)
            assert not w[2 * (2 * nt - 1):].any()                          # nothing beyond the words the vertical pass reads
    for bad in (0, 2, 4, 33):
        with pytest.raises(api.BsxError):
            api.gauss_coeff_words(bad, 0)


def test_bilateral_rounding_tie_audit(oracle):
    """Decision-margin audit of the bilateral's final rounding (OpenCV absent: unpinned).  out = cvRound(sum * (1 / wsum)) in f32; an OpenCV build that contracts the
    accumulation into FMAs (or sums in another order) moves `sum` by a few ulp, which changes the 8-bit result only where sum / wsum sits within a few ulp of x.5.
    Count those values on the canvases the models really see (photo fixture + synthetic scene, resized as prep does): the upper bound on network-input bytes that can
    differ by 1 LSB from the real reference.  Recorded in DESIGN.md §2."""
    from backscrub_amd import synth
    from tools import make_photo_fixture
    lut = np.exp(np.arange(768, dtype=np.float64) ** 2 * (-0.5 / 100.0 ** 2)).astype(np.float32)
    taps = [(i, j) for i in range(-2, 3) for j in range(-2, 3) if np.sqrt(i * i + j * j) <= 2]
    report = {}
    for name, frame in (("photo", make_photo_fixture.load_frames()[0]), ("synthetic", synth.frame(640, 480, 0, 0))):
        for tag, (dw, dh), roi in (("mlkit 256x256", (256, 256), (80, 0, 480, 480)), ("meet-lite 128x96", (128, 96), (0, 0, 640, 480))):
            x0, y0, w, h = roi
            img = oracle.resize_linear(np.ascontiguousarray(frame[y0:y0 + h, x0:x0 + w]), dw, dh)[..., ::-1].copy()      # BGR2RGB
            p = np.pad(img, ((2, 2), (2, 2), (0, 0)), mode="reflect").astype(np.int32)
            c = img.astype(np.int32)
            acc = np.zeros(img.shape, np.float32)
            ws = np.zeros(img.shape[:2], np.float32)
            for i, j in taps:
                nb = p[2 + i:2 + i + dh, 2 + j:2 + j + dw]
                wgt = np.float32(np.exp((i * i + j * j) * (-0.5 / 100.0 ** 2))) * lut[np.abs(nb - c).sum(-1)]
                acc = acc + nb.astype(np.float32) * wgt[..., None]
                ws = ws + wgt
            v = acc * (np.float32(1) / ws)[..., None]
            assert np.array_equal(np.rint(v).astype(np.uint8), oracle.bilateral(img))
            frac = np.abs(v - np.floor(v) - np.float32(0.5))
            ulp = np.spacing(np.maximum(v, np.float32(1)).astype(np.float32))
            near = int((frac <= 4 * ulp).sum())                  # 13 accumulations: an FMA / reordered build stays within ~4 ulp of this one
            report["%s, %s" % (name, tag)] = (near, int(v.size))
    print("bilateral tie audit (values within 4 ulp of x.5 / all values):", report)
    for near, total in report.values():
        assert near <= 2e-4 * total
