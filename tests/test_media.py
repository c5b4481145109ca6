"""Background media decoders (media.cpp; SURVEY §8 f4 — what cv::imread / cv::VideoCapture hand to load_background,
/root/reference/app/background.cc:126-176).  Host-only: the checker is Pillow's decoder on the same bytes."""
import io
import os
import struct

import numpy as np
import pytest

import backscrub_amd

Image = pytest.importorskip("PIL.Image")
from PIL import ImageSequence  # noqa: E402


def _pil_frames(path):
    im = Image.open(path)
    return np.stack([np.asarray(f.convert("RGB"))[:, :, ::-1] for f in ImageSequence.Iterator(im)])


def _rand_rgb(rng, h, w):
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([(x * 5 + y) % 256, (y * 7) % 256, (x ^ y) % 256], -1).astype(np.uint8)
    base[rng.integers(0, h, 40), rng.integers(0, w, 40)] = rng.integers(0, 256, (40, 3), dtype=np.uint8)
    return base


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA", "P"])
@pytest.mark.parametrize("size", [(1, 1), (37, 23), (640, 480)])
def test_png_matches_pillow(tmp_path, mode, size):
    rng = np.random.default_rng(5)
    w, h = size
    im = Image.fromarray(_rand_rgb(rng, h, w), "RGB")
    if mode == "P":
        im = im.quantize(64)
    elif mode in ("RGBA", "LA"):
        im = im.convert(mode)
        im.putalpha(Image.fromarray(rng.integers(0, 256, (h, w), dtype=np.uint8), "L"))
    else:
        im = im.convert(mode)
    path = tmp_path / "a.png"
    im.save(path)
    frames, fps = backscrub_amd.media_decode(str(path))
    assert fps == 0 and frames.shape == (1, h, w, 3)
    # cv::imread(IMREAD_COLOR) drops alpha and expands grey / palette to BGR
    want = np.asarray(Image.open(path).convert("RGBA").convert("RGB") if mode in ("RGBA", "LA") else Image.open(path).convert("RGB"))[:, :, ::-1]
    assert np.array_equal(frames[0], want)


def test_png_of_the_committed_photo_fixture():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "photo_2x640x480.png")
    frames, _ = backscrub_amd.media_decode(path)
    want = np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1]
    assert np.array_equal(frames[0], want)


def test_ppm(tmp_path):
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (9, 13, 3), dtype=np.uint8)
    p = tmp_path / "a.ppm"
    p.write_bytes(b"P6\n# comment\n13 9\n255\n" + img.tobytes())
    frames, _ = backscrub_amd.media_decode(str(p))
    assert np.array_equal(frames[0], img[:, :, ::-1])


@pytest.mark.parametrize("n,size,dur", [(1, (33, 17), 100), (5, (64, 48), 40), (12, (160, 120), 100)])
def test_gif_animation_matches_pillow(tmp_path, n, size, dur):
    rng = np.random.default_rng(n)
    w, h = size
    ims = []
    for i in range(n):
        a = _rand_rgb(rng, h, w)
        a = np.roll(a, 3 * i, axis=1)
        ims.append(Image.fromarray(a, "RGB").quantize(128))
    path = tmp_path / "a.gif"
    ims[0].save(path, save_all=n > 1, append_images=ims[1:], duration=dur, loop=0, optimize=False)
    frames, fps = backscrub_amd.media_decode(str(path))
    want = _pil_frames(path)
    assert frames.shape == want.shape
    assert np.array_equal(frames, want)
    if n > 1:
        assert abs(fps - 1000.0 / dur) < 1e-6


# ---- a GIF writer for what Pillow cannot write: interlaced images, sub-rectangles, disposal methods, transparency ----
def _lzw_plain(indices, mcs=8):
    """'uncompressed' LZW: a clear code often enough that the code size never grows"""
    clear, eoi = 1 << mcs, (1 << mcs) + 1
    codes = []
    for i, v in enumerate(indices):
        if i % ((1 << mcs) - 2) == 0:
            codes.append(clear)
        codes.append(int(v))
    codes.append(eoi)
    bits, nb, out = 0, 0, bytearray()
    for c in codes:
        bits |= c << nb
        nb += mcs + 1
        while nb >= 8:
            out.append(bits & 255)
            bits >>= 8
            nb -= 8
    if nb:
        out.append(bits & 255)
    blocks = bytearray([mcs])
    for i in range(0, len(out), 255):
        chunk = out[i:i + 255]
        blocks.append(len(chunk))
        blocks += chunk
    blocks.append(0)
    return bytes(blocks)


def _interlace_rows(h):
    return [r for s, st in ((0, 8), (4, 8), (2, 4), (1, 2)) for r in range(s, h, st)]


def _gif(w, h, palette, frames):
    """frames: dicts(x, y, idx[h',w'], interlace, disposal, transparent, delay_cs)"""
    b = bytearray(b"GIF89a" + struct.pack("<HHBBB", w, h, 0xF7, 0, 0) + bytes(palette))
    for f in frames:
        idx = np.asarray(f["idx"], np.uint8)
        fh, fw = idx.shape
        t = f.get("transparent")
        b += bytes([0x21, 0xF9, 4, (f.get("disposal", 0) << 2) | (1 if t is not None else 0)]) + struct.pack("<H", f.get("delay_cs", 10)) + bytes([t or 0, 0])
        b += b"," + struct.pack("<HHHHB", f.get("x", 0), f.get("y", 0), fw, fh, 0x40 if f.get("interlace") else 0)
        rows = idx[_interlace_rows(fh)] if f.get("interlace") else idx
        b += _lzw_plain(rows.reshape(-1))
    b += b";"
    return bytes(b)


def test_gif_interlace_subrect_disposal_transparency(tmp_path):
    rng = np.random.default_rng(3)
    pal = rng.integers(0, 256, (256, 3), dtype=np.uint8)
    pal[0] = (10, 20, 30)
    W, H = 40, 30
    frames = [
        dict(idx=rng.integers(1, 256, (H, W)), interlace=True, disposal=1, delay_cs=5),
        dict(x=5, y=3, idx=rng.integers(1, 256, (11, 17)), disposal=2, delay_cs=5),                          # restore to background afterwards
        dict(x=20, y=10, idx=rng.integers(1, 256, (13, 9)), interlace=True, disposal=3, delay_cs=5),          # restore to previous afterwards
        dict(x=0, y=0, idx=rng.integers(1, 256, (H, W // 2)), transparent=200, disposal=1, delay_cs=5),
        dict(x=2, y=2, idx=rng.integers(1, 256, (5, 5)), delay_cs=5),
    ]
    frames[3]["idx"][::2, ::3] = 200
    path = tmp_path / "h.gif"
    path.write_bytes(_gif(W, H, pal.tobytes(), frames))
    got, fps = backscrub_amd.media_decode(str(path))
    want = _pil_frames(path)
    assert got.shape == want.shape == (5, H, W, 3)
    for i in range(5):
        assert np.array_equal(got[i], want[i]), "frame %d" % i
    assert abs(fps - 20.0) < 1e-9


def test_gif_real_lzw_with_growing_codes(tmp_path):
    # Pillow's encoder on smooth content exercises code-size growth, table reset at 4096 entries and the KwKwK case
    y, x = np.mgrid[0:200, 0:300]
    idx = ((x // 3 + y // 2) % 256).astype(np.uint8)
    idx[50:150, 100:200] = 9
    im = Image.fromarray(idx, "P")
    im.putpalette(np.random.default_rng(0).integers(0, 256, 768, dtype=np.uint8).tobytes())
    path = tmp_path / "g.gif"
    im.save(path)
    got, _ = backscrub_amd.media_decode(str(path))
    assert np.array_equal(got, _pil_frames(path))


def test_gif_restore_to_background_of_a_frame_with_transparency_follows_libavcodec(tmp_path):
    # where decoders disagree (Pillow fills with the transparent index's colour): cv::VideoCapture's GIF decoder is libavcodec's, which
    # fills with its TRANSPARENT colour — 0x00ffffff, white once BGRA→BGR drops the alpha; and a first frame smaller than the screen leaves
    # background colour around it
    rng = np.random.default_rng(4)
    pal = rng.integers(1, 256, (256, 3), dtype=np.uint8)
    W, H = 16, 12
    f0 = rng.integers(1, 256, (6, 8))
    f1 = rng.integers(1, 256, (4, 4))
    f1[1, 1] = 9
    f2 = rng.integers(1, 256, (2, 2))
    path = tmp_path / "t.gif"
    path.write_bytes(_gif(W, H, pal.tobytes(), [dict(x=2, y=1, idx=f0, disposal=1), dict(x=3, y=2, idx=f1, transparent=9, disposal=2), dict(x=0, y=0, idx=f2)]))
    got, _ = backscrub_amd.media_decode(str(path))
    bgr = pal[:, ::-1]
    c = np.empty((H, W, 3), np.uint8)
    c[:] = bgr[0]                                        # background colour index 0 (no transparency on the first image)
    c[1:7, 2:10] = bgr[f0]
    assert np.array_equal(got[0], c)
    keep = c[3, 4].copy()
    c[2:6, 3:7] = bgr[f1]
    c[3, 4] = keep                                        # transparent pixel shows what was there
    assert np.array_equal(got[1], c)
    c[2:6, 3:7] = 255                                     # restored to "background" = transparent = libavcodec's transparent white
    c[0:2, 0:2] = bgr[f2]
    assert np.array_equal(got[2], c)


# ---- JPEG (jpeg.cpp): the checker is Pillow's libjpeg-turbo — the library cv::imread decodes JPEG with — on the same bytes ----
def _photo_like(rng, h, w):
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    a = np.stack([128 + 100 * np.sin(x / 9.0 + y / 17.0), 128 + 90 * np.cos(x / 5.0) * np.sin(y / 7.0), (x * 3 + y * 2) % 256], -1)
    a += rng.normal(0, 12, a.shape)
    a[h // 3: h // 2, w // 4: w // 2] = (250, 10, 30)                 # saturated patch: hard chroma edges through the up-sampling filter
    return np.clip(a, 0, 255).astype(np.uint8)


@pytest.mark.parametrize("progressive", [False, True])
@pytest.mark.parametrize("subsampling", ["4:4:4", "4:2:2", "4:2:0"])
@pytest.mark.parametrize("size", [(1, 1), (3, 2), (5, 4), (17, 13), (64, 48), (161, 97)])
def test_jpeg_matches_libjpeg(tmp_path, progressive, subsampling, size):
    rng = np.random.default_rng(size[0] * 7 + size[1])
    w, h = size
    path = tmp_path / "a.jpg"
    Image.fromarray(_photo_like(rng, h, w), "RGB").save(path, "JPEG", quality=int(rng.integers(30, 96)), subsampling=subsampling, progressive=progressive, optimize=bool(w & 1))
    frames, fps = backscrub_amd.media_decode(str(path))
    want = np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1]
    assert fps == 0 and frames.shape == (1, h, w, 3)
    assert np.array_equal(frames[0], want)


@pytest.mark.parametrize("kw", [dict(quality=100, subsampling="4:4:4"), dict(quality=3), dict(quality=75, restart_marker_blocks=3), dict(quality=60, restart_marker_rows=1, progressive=True),
                                dict(quality=85, keep_rgb=True), dict(quality=50, qtables="web_high")])
def test_jpeg_encoder_variants(tmp_path, kw):
    rng = np.random.default_rng(21)
    path = tmp_path / "v.jpg"
    Image.fromarray(_photo_like(rng, 75, 131), "RGB").save(path, "JPEG", **kw)
    frames, _ = backscrub_amd.media_decode(str(path))
    assert np.array_equal(frames[0], np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])


@pytest.mark.parametrize("progressive", [False, True])
def test_jpeg_greyscale_expands_to_bgr(tmp_path, progressive):
    rng = np.random.default_rng(2)
    path = tmp_path / "g.jpg"
    Image.fromarray(_photo_like(rng, 50, 70)[:, :, 0], "L").save(path, "JPEG", quality=80, progressive=progressive)
    frames, _ = backscrub_amd.media_decode(str(path))
    want = np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1]                 # cv::imread(IMREAD_COLOR): grey replicated
    assert np.array_equal(frames[0], want)


@pytest.mark.parametrize("orientation", range(1, 9))
def test_jpeg_exif_orientation_is_applied_as_imread_does(tmp_path, orientation):
    from PIL import ImageOps
    rng = np.random.default_rng(orientation)
    im = Image.fromarray(_photo_like(rng, 40, 56), "RGB")
    exif = Image.Exif()
    exif[0x0112] = orientation
    path = tmp_path / "o.jpg"
    im.save(path, "JPEG", quality=90, exif=exif)
    frames, _ = backscrub_amd.media_decode(str(path))
    want = np.asarray(ImageOps.exif_transpose(Image.open(path)).convert("RGB"))[:, :, ::-1]
    assert frames.shape[1:] == want.shape
    assert np.array_equal(frames[0], want)


def test_jpeg_reference_style_background(tmp_path):
    """the shape of the reference's own `backgrounds/total_landscaping.jpg`: 1280x720, progressive, 4:2:0, JFIF"""
    rng = np.random.default_rng(8)
    path = tmp_path / "bg.jpg"
    Image.fromarray(_photo_like(rng, 720, 1280), "RGB").save(path, "JPEG", quality=82, subsampling="4:2:0", progressive=True)
    frames, _ = backscrub_amd.media_decode(str(path))
    assert np.array_equal(frames[0], np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])


@pytest.mark.parametrize("name", ["total_landscaping.jpg", "screenshot.jpg"])
def test_jpeg_the_reference_backgrounds_where_present(name):
    path = os.path.join("/root/reference/backgrounds", name)
    if not os.path.exists(path):
        pytest.skip("reference checkout not present on this machine")
    frames, _ = backscrub_amd.media_decode(path)
    assert np.array_equal(frames[0], np.asarray(Image.open(path).convert("RGB"))[:, :, ::-1])


def test_jpeg_unsupported_processes_are_refused(tmp_path):
    rng = np.random.default_rng(1)
    b = io.BytesIO()
    Image.fromarray(_photo_like(rng, 16, 16), "RGB").convert("CMYK").save(b, "JPEG")
    p = tmp_path / "cmyk.jpg"
    p.write_bytes(b.getvalue())
    with pytest.raises(backscrub_amd.BsxError, match="CMYK"):
        backscrub_amd.media_decode(str(p))
    b = io.BytesIO()
    Image.fromarray(_photo_like(rng, 16, 16), "RGB").save(b, "JPEG")
    a = bytearray(b.getvalue())
    i = a.find(b"\xff\xc0")
    a[i + 1] = 0xC9                                          # arithmetic-coded sequential DCT
    p.write_bytes(bytes(a))
    with pytest.raises(backscrub_amd.BsxError, match="arithmetic"):
        backscrub_amd.media_decode(str(p))


@pytest.mark.parametrize("blob", [b"", b"GIF89a", b"\x89PNG\r\n\x1a\n", b"P6\n1 1\n255\n", b"\xff\xd8\xff\xe0JFIF", b"RIFF....WEBP", b"\x1a\x45\xdf\xa3"])
def test_truncated_and_unsupported_files_are_rejected(tmp_path, blob):
    p = tmp_path / "bad.bin"
    p.write_bytes(blob)
    with pytest.raises(backscrub_amd.BsxError):
        backscrub_amd.media_decode(str(p))


def test_corrupted_files_never_crash(tmp_path):
    rng = np.random.default_rng(11)
    im = Image.fromarray(_rand_rgb(rng, 40, 50), "RGB")
    bufs = []
    for fmt, kw in (("PNG", {}), ("GIF", {}), ("JPEG", {}), ("JPEG", dict(progressive=True, subsampling="4:2:0")), ("JPEG", dict(restart_marker_blocks=2))):
        b = io.BytesIO()
        (im.quantize(64) if fmt == "GIF" else im).save(b, fmt, **kw)
        bufs.append(b.getvalue())
    p = tmp_path / "c.bin"
    for base in bufs:
        for trial in range(150):
            a = bytearray(base)
            for _ in range(rng.integers(1, 6)):
                a[rng.integers(0, len(a))] = rng.integers(0, 256)
            if trial % 3 == 0:
                a = a[: rng.integers(1, len(a))]
            p.write_bytes(bytes(a))
            try:
                frames, _ = backscrub_amd.media_decode(str(p))
                assert frames.ndim == 4 and frames.shape[0] >= 1
            except backscrub_amd.BsxError:
                pass


def test_missing_file():
    with pytest.raises(backscrub_amd.BsxError):
        backscrub_amd.media_decode("/nonexistent/bg.png")
