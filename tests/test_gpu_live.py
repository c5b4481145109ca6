"""SURVEY §8 f4 on the GPU: the background source (/root/reference/app/background.cc) and the CalcMask worker
(/root/reference/app/deepseg.cc:159-286), through the C ABI."""
import time

import numpy as np
import pytest

from conftest import model_path, synthetic_model_path

pytestmark = pytest.mark.gpu
VGA = (640, 480)


@pytest.fixture(scope="module")
def bs():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU test selected but no GPU visible")
    import backscrub_amd
    return backscrub_amd


def test_still_background_grab_is_the_resized_image(bs, oracle, tmp_path):
    from PIL import Image
    from backscrub_amd import synth
    img = synth.random_u8((333, 517, 3), 3)
    p = tmp_path / "bg.png"
    Image.fromarray(img[:, :, ::-1].copy(), "RGB").save(p)
    mg = bs.MaskGen(synthetic_model_path("lite"), *VGA, n_streams=1)
    bg = bs.Background(mg, path=str(p))
    assert (bg.width, bg.height, bg.n_frames, bg.video) == (517, 333, 1, False)
    for size in (VGA, (1280, 720), (517, 333)):
        frm, out = bg.grab(*size)
        assert frm == 1                                                  # background.cc:190-192: a still image reports frame 1
        assert np.array_equal(out.cpu().numpy(), oracle.resize_linear(img, *size))
    bg.close()
    with pytest.raises(bs.BsxError):
        bs.Background(mg, path=str(tmp_path / "missing.png"))
    (tmp_path / "x.jpg").write_bytes(b"\xff\xd8\xff\xe0JFIF")
    with pytest.raises(bs.BsxError):
        bs.Background(mg, path=str(tmp_path / "x.jpg"))
    mg.close()


def test_animated_background_advances_at_its_fps_and_loops(bs, oracle, tmp_path):
    from PIL import Image
    from backscrub_amd import synth
    n, dur_ms = 6, 40
    frames = synth.random_u8((n, 90, 120, 3), 8) // 64 * 64            # 64 colours: fits one GIF palette
    ims = [Image.fromarray(f[:, :, ::-1].copy(), "RGB").quantize(256, dither=Image.Dither.NONE) for f in frames]
    p = tmp_path / "bg.gif"
    ims[0].save(p, save_all=True, append_images=ims[1:], duration=dur_ms, loop=0, optimize=False)
    decoded, fps = bs.media_decode(str(p))
    assert decoded.shape == frames.shape and abs(fps - 25.0) < 1e-6
    mg = bs.MaskGen(synthetic_model_path("lite"), *VGA, n_streams=1)
    bg = bs.Background(mg, path=str(p))
    assert bg.video and bg.n_frames == n
    seen, t0 = [], time.time()
    while time.time() - t0 < 0.7:                                       # ~17 frame periods: at least two trips round the loop
        frm, out = bg.grab(*VGA)
        assert 1 <= frm <= n                                            # the reference counts pictures READ: picture c is reported as c + 1 (background.cc:60-63)
        assert np.array_equal(out.cpu().numpy(), oracle.resize_linear(decoded[frm - 1], *VGA)), frm
        seen.append(frm - 1)
        time.sleep(0.005)
    steps = [(b - a) % n for a, b in zip(seen, seen[1:])]
    assert set(steps) <= {0, 1, 2}                                      # paced: never jumps ahead
    assert set(seen) == set(range(n))                                   # every frame shown
    assert any(b < a for a, b in zip(seen, seen[1:]))                   # wrapped to frame 0 (background.cc:91-95)
    changes = sum(1 for s in steps if s)
    assert 12 <= changes <= 22, changes                                 # ≈ 0.7 s x 25 fps
    bg.close()
    # caller-decoded frames (the boundary for codecs this library does not carry)
    bg = bs.Background(mg, frames=decoded, fps=0.0)
    assert not bg.video and bg.grab(*VGA)[0] == 1
    bg.close()
    mg.close()


@pytest.mark.parametrize("key", ["lite", "deeplab"])
def test_live_worker_masks_match_the_oracle_frame_by_frame(bs, oracle, key):
    from backscrub_amd import synth
    W, H = VGA
    path = model_path(key)
    mg = bs.MaskGen(path, W, H, n_streams=1)
    oc = oracle.Ctx(path, W, H)
    live = bs.Live(mg)
    mask = np.full((H, W), 7, np.uint8)
    assert live.get_output_mask(mask) is False and (mask == 7).all()     # nothing yet: the caller's mask is left alone (deepseg.cc:279-285)
    for t in range(4):
        f = synth.frame(W, H, 0, t)
        live.set_input_frame(f)
        f[:] = 0                                                         # the worker owns a copy (frame.clone(), :273)
        t0 = time.time()
        while not live.get_output_mask(mask):
            assert time.time() - t0 < 20, "worker never produced a mask"
            time.sleep(0.0005)
        want = oc.process(synth.frame(W, H, 0, t))
        inter = ((mask < 128) & (want < 128)).sum()
        union = ((mask < 128) | (want < 128)).sum()
        assert union == 0 or inter / union >= 0.999
        assert (np.abs(mask.astype(int) - want.astype(int)) > 1).mean() <= 1e-3
        assert live.get_output_mask(mask) is False                       # consumed
    live.close()
    oc.close()
    mg.close()


def test_live_worker_never_blocks_the_camera_loop(bs):
    from backscrub_amd import synth
    W, H = VGA
    mg = bs.MaskGen(model_path("lite"), W, H, n_streams=1)
    live = bs.Live(mg)
    mask = np.full((H, W), 255, np.uint8)
    got, t0 = 0, time.time()
    for t in range(200):                                                 # faster than the worker can follow: frames are dropped, not queued
        live.set_input_frame(synth.frame(W, H, 0, t % 4))
        got += live.get_output_mask(mask)
    dt = time.time() - t0
    live.close()                                                         # drains the queue
    mg.close()
    assert got >= 1 and dt < 20
