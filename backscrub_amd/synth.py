"""Seeded synthetic camera frames / backgrounds.

No datasets or camera exist in the build or bench environment, so every test and bench
input comes from here: a smooth gradient room, a centred soft "person" (head ellipse +
torso) in skin/cloth tones, and ±8 uniform sensor noise so the bilateral filter and the
mask boundary both matter.  Pure-random frames are available for integer-kernel stress.
"""
from __future__ import annotations

import numpy as np

SEED_BASE = 0xB5C0


def _ell(xx, yy, cx, cy, rx, ry, soft=6.0):
    d = ((xx - cx) / rx) ** 2 + ((yy - cy) / ry) ** 2
    return np.clip((1.0 - d) * soft, 0, 1)[..., None]


def frame(width: int, height: int, stream: int = 0, t: int = 0, noise: int = 6) -> np.ndarray:
    """BGR u8 [H,W,3]; `stream` picks the scene (wall colours, person position/size/clothes),
    `t` sways the person a little and reseeds the sensor noise.  The figure (hair, face with
    eyes/mouth, neck, dark-clothed shoulders on a light wall) is person-like enough that the
    real Meet and MLKit networks segment it (checked against the CPU oracle)."""
    rng = np.random.default_rng(SEED_BASE + stream * 1009 + t)
    srng = np.random.default_rng(SEED_BASE + stream)
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    u, v = xx / width, yy / height
    wall = srng.uniform(170, 215, 3)
    img = np.stack([wall[k] - 60 * v + 20 * np.sin(5 * u + k + stream) for k in range(3)], -1)
    cx = (0.5 + srng.uniform(-0.1, 0.1) + 0.01 * np.sin(0.7 * t)) * width
    cy = (0.42 + srng.uniform(-0.04, 0.04)) * height
    rh = srng.uniform(0.15, 0.19) * height

    def put(im, a, col):
        return im * (1 - a) + np.asarray(col, np.float32) * a

    cloth = srng.uniform(25, 90, 3)
    skin = np.array([125, 155, 215], np.float32) + srng.uniform(-12, 12, 3)
    img = put(img, _ell(xx, yy, cx, cy + 3.1 * rh, 2.6 * rh, 2.0 * rh), cloth)              # shoulders
    img = put(img, _ell(xx, yy, cx, cy + 1.1 * rh, 0.45 * rh, 0.6 * rh), skin * 0.9)         # neck
    img = put(img, _ell(xx, yy, cx, cy - 0.15 * rh, 0.88 * rh, 1.05 * rh), (30, 35, 45))     # hair
    img = put(img, _ell(xx, yy, cx, cy + 0.1 * rh, 0.72 * rh, 0.92 * rh), skin)              # face
    for sx in (-0.3, 0.3):
        img = put(img, _ell(xx, yy, cx + sx * rh, cy - 0.05 * rh, 0.11 * rh, 0.06 * rh, 8), (40, 40, 50))
    img = put(img, _ell(xx, yy, cx, cy + 0.5 * rh, 0.25 * rh, 0.07 * rh, 8), (70, 70, 150))  # mouth
    img = put(img, _ell(xx, yy, cx, cy + 0.2 * rh, 0.07 * rh, 0.18 * rh, 3) * 0.4, (90, 120, 180))
    if noise:
        img = img + rng.integers(-noise, noise + 1, img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def frames(n: int, width: int, height: int, t: int = 0, distinct: int | None = None) -> np.ndarray:
    """[n,H,W,3]; only `distinct` different scenes are rendered (rest are re-noised copies)."""
    distinct = min(n, distinct or n)
    base = [frame(width, height, s, t) for s in range(distinct)]
    out = np.empty((n, height, width, 3), np.uint8)
    for i in range(n):
        if i < distinct:
            out[i] = base[i]
        else:
            rng = np.random.default_rng(SEED_BASE + 7919 * i + t)
            out[i] = np.clip(base[i % distinct].astype(np.int16) + rng.integers(-3, 4, base[0].shape, dtype=np.int16), 0, 255)
    return out


def background(width: int, height: int, seed: int = 1) -> np.ndarray:
    rng = np.random.default_rng(SEED_BASE ^ (seed * 7717))
    yy, xx = np.mgrid[0:height, 0:width].astype(np.float32)
    img = np.stack([127 + 100 * np.sin(xx / width * rng.uniform(3, 9) + k) * np.cos(yy / height * rng.uniform(3, 9) - k)
                    for k in range(3)], -1)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def random_u8(shape, seed=0) -> np.ndarray:
    return np.random.default_rng(SEED_BASE + seed).integers(0, 256, shape, dtype=np.uint8)
