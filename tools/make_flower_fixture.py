#!/usr/bin/env python
"""Generate the bundled sample dataset ``recognition_flower/flower_photos_labeled/``.

The reference ships a 107-image, 5-class sample of ``flower_photos`` next to its scripts and uses it as the
default ``--dataset_path`` (recognition_flower/single/flower_train.py:22-23,
recognition_flower/distributed/prepare.py:10-11).  Photographs cannot be redistributed from here, so this tool
*draws* a stand-in with the same shape: the same five class directories with the same number of files each
(daisy 21, dandelion 24, roses 21, sunflower 21, tulips 20), baseline 4:4:4 RGB JPEGs, widths 175-500 and
heights 212-399 with the reference's most common sizes (500x375, 320x240, 500x333, 180x240) over-represented.
Every class has its own petal count / shape / colour family, with random position, size, rotation, background
and noise, so that a classifier can actually learn the five classes.  The output is a pure function of
``--seed``.

    python tools/make_flower_fixture.py [--out recognition_flower/flower_photos_labeled] [--seed 20181]
"""
from __future__ import annotations

import argparse
import os

import numpy as np

CLASSES = {"daisy": 21, "dandelion": 24, "roses": 21, "sunflower": 21, "tulips": 20}
COMMON_SIZES = [(500, 375), (320, 240), (500, 333), (180, 240)]  # (width, height)


def _size(rng):
    if rng.random() < 0.6:
        return COMMON_SIZES[int(rng.integers(len(COMMON_SIZES)))]
    return int(rng.integers(175, 501)), int(rng.integers(212, 400))


def _background(rng, h, w, base):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    img = np.empty((h, w, 3), np.float32)
    g = (yy / h)[..., None]
    top = np.asarray(base, np.float32) + rng.normal(0, 12, 3)
    bot = np.asarray(base, np.float32) * 0.6 + rng.normal(0, 12, 3)
    img[:] = top * (1 - g) + bot * g
    # soft blotches (out-of-focus foliage)
    for _ in range(int(rng.integers(4, 10))):
        cx, cy, r = rng.uniform(0, w), rng.uniform(0, h), rng.uniform(0.08, 0.3) * max(h, w)
        blot = np.exp(-(((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r)))[..., None]
        img += blot * rng.normal(0, 25, 3)
    return img


def _polar(h, w, cx, cy, rot):
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    dx, dy = xx - cx, yy - cy
    return np.sqrt(dx * dx + dy * dy), np.arctan2(dy, dx) + rot, dx, dy


def _blend(img, mask, colour, shade=None):
    m = np.clip(mask, 0, 1)[..., None]
    c = np.asarray(colour, np.float32)
    if shade is not None:
        c = c * shade[..., None]
    img[:] = img * (1 - m) + c * m


def _rosette(img, rng, cx, cy, R, n, petal, centre, width_pow, centre_frac, inner=0.25):
    """n petals: radius profile R * (inner + (1 - inner) * |cos(n t / 2)|^width_pow), a disc of centre_frac * R."""
    h, w = img.shape[:2]
    r, t, _, _ = _polar(h, w, cx, cy, rng.uniform(0, np.pi))
    prof = R * (inner + (1 - inner) * np.abs(np.cos(n * t / 2)) ** width_pow)
    edge = max(1.0, R * 0.03)
    shade = 0.75 + 0.25 * np.clip(r / max(R, 1), 0, 1)
    _blend(img, (prof - r) / edge, petal, shade)
    _blend(img, (centre_frac * R - r) / edge, centre)


def draw_daisy(img, rng):
    h, w = img.shape[:2]
    for _ in range(int(rng.integers(1, 3))):
        R = rng.uniform(0.22, 0.42) * min(h, w)
        _rosette(img, rng, rng.uniform(0.3, 0.7) * w, rng.uniform(0.3, 0.7) * h, R, int(rng.integers(16, 25)),
                 (245, 245, 238) + rng.normal(0, 5, 3), (235, 190, 30) + rng.normal(0, 10, 3), 0.45, 0.22)


def draw_dandelion(img, rng):
    h, w = img.shape[:2]
    for _ in range(int(rng.integers(1, 3))):
        R = rng.uniform(0.2, 0.38) * min(h, w)
        cx, cy = rng.uniform(0.3, 0.7) * w, rng.uniform(0.3, 0.7) * h
        # many thin ray florets in two layers
        _rosette(img, rng, cx, cy, R, int(rng.integers(44, 64)), (250, 215, 20) + rng.normal(0, 8, 3),
                 (240, 180, 10) + rng.normal(0, 8, 3), 1.6, 0.12, inner=0.55)
        _rosette(img, rng, cx, cy, R * 0.7, int(rng.integers(30, 44)), (255, 225, 40) + rng.normal(0, 8, 3),
                 (235, 170, 10) + rng.normal(0, 8, 3), 1.2, 0.15, inner=0.6)


def draw_rose(img, rng):
    h, w = img.shape[:2]
    base = np.asarray([(200, 25, 45), (230, 90, 130), (180, 20, 90)][int(rng.integers(3))], np.float32)
    for _ in range(int(rng.integers(1, 3))):
        R = rng.uniform(0.2, 0.36) * min(h, w)
        cx, cy = rng.uniform(0.3, 0.7) * w, rng.uniform(0.3, 0.7) * h
        # concentric layers of a few wide overlapping petals, darker towards the heart
        for k, frac in enumerate((1.0, 0.78, 0.58, 0.4, 0.24)):
            col = base * (1.0 - 0.12 * k) + rng.normal(0, 6, 3)
            _rosette(img, rng, cx, cy, R * frac, int(rng.integers(5, 8)), col, col * 0.8, 0.25, 0.0, inner=0.7)


def draw_sunflower(img, rng):
    h, w = img.shape[:2]
    R = rng.uniform(0.3, 0.46) * min(h, w)
    cx, cy = rng.uniform(0.35, 0.65) * w, rng.uniform(0.35, 0.65) * h
    _rosette(img, rng, cx, cy, R, int(rng.integers(14, 21)), (250, 190, 15) + rng.normal(0, 8, 3),
             (70, 40, 15) + rng.normal(0, 6, 3), 0.6, 0.45, inner=0.5)
    # seed pattern in the disc
    r, t, _, _ = _polar(h, w, cx, cy, 0.0)
    disc = (r < 0.43 * R)
    tex = 0.5 + 0.5 * np.sin(r * 0.9 + 7 * t)
    img[disc] *= (0.8 + 0.25 * tex[disc])[..., None]


def draw_tulip(img, rng):
    h, w = img.shape[:2]
    for _ in range(int(rng.integers(1, 4))):
        col = np.asarray([(215, 30, 40), (150, 50, 170), (245, 130, 20), (240, 210, 60)][int(rng.integers(4))],
                         np.float32) + rng.normal(0, 8, 3)
        H = rng.uniform(0.28, 0.42) * h
        W = H * rng.uniform(0.55, 0.75)
        cx, cy = rng.uniform(0.2, 0.8) * w, rng.uniform(0.3, 0.5) * h
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        # stem
        stem = (np.abs(xx - cx - 0.05 * W * np.sin((yy - cy) / H * 3)) < max(2.0, 0.04 * W)) & (yy > cy)
        img[stem] = np.asarray((60, 130, 50), np.float32) + rng.normal(0, 5, 3)
        # cup: three upright petals = egg shape with a notched top
        u, v = (xx - cx) / (W / 2), (yy - cy) / (H / 2)
        egg = u * u / (1.0 - 0.35 * v) ** 2 + v * v
        notch = 0.22 * np.abs(np.cos(1.5 * np.pi * u)) * (v < 0)
        mask = (1.0 - egg - notch * 1.6) * 12
        shade = 0.8 + 0.2 * np.cos(np.pi * u * 1.5) ** 2
        _blend(img, mask, col, shade)


DRAW = {"daisy": (draw_daisy, (70, 125, 60)), "dandelion": (draw_dandelion, (50, 105, 45)),
        "roses": (draw_rose, (45, 85, 50)), "sunflower": (draw_sunflower, (110, 160, 215)),
        "tulips": (draw_tulip, (150, 185, 140))}


def make_image(cls: str, rng: np.random.Generator) -> np.ndarray:
    w, h = _size(rng)
    fn, bg = DRAW[cls]
    img = _background(rng, h, w, bg)
    fn(img, rng)
    img += rng.normal(0, 4.0, img.shape)  # sensor noise keeps the files from being trivially compressible
    return np.clip(img, 0, 255).astype(np.uint8)


def generate(out_dir: str, seed: int = 20181, quality: int = 88) -> int:
    from PIL import Image

    n = 0
    for ci, (cls, count) in enumerate(CLASSES.items()):
        d = os.path.join(out_dir, cls)
        os.makedirs(d, exist_ok=True)
        for i in range(count):
            rng = np.random.default_rng([seed, ci, i])
            img = make_image(cls, rng)
            # file names in the style of flower_photos: <numeric id>_<hex>.jpg
            name = "%d_%010x.jpg" % (int(rng.integers(10 ** 8, 10 ** 10)), int(rng.integers(16 ** 10)))
            Image.fromarray(img).save(os.path.join(d, name), format="JPEG", quality=quality, subsampling=0,
                                      optimize=False, progressive=False)
            n += 1
    return n


def main():
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(here, "recognition_flower", "flower_photos_labeled"))
    ap.add_argument("--seed", type=int, default=20181)
    ap.add_argument("--quality", type=int, default=88)
    a = ap.parse_args()
    n = generate(a.out, a.seed, a.quality)
    total = sum(os.path.getsize(os.path.join(r, f)) for r, _, fs in os.walk(a.out) for f in fs)
    print("wrote %d images (%.1f MB) under %s" % (n, total / 1e6, a.out))


if __name__ == "__main__":
    main()
