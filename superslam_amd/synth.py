"""Procedural KITTI-shaped synthetic frames (no datasets offline).

Multi-octave value noise + random filled rectangles + a 3x3 binomial blur; the right image is
the left one shifted by a per-row-band disparity so a matcher has real correspondences.
Pure numpy with a seeded PCG64 generator: bit-reproducible here and on the GPU box.
"""
from __future__ import annotations

import numpy as np


def _value_noise(rng, h, w, cell):
    gh, gw = h // cell + 2, w // cell + 2
    g = rng.random((gh, gw), dtype=np.float32)
    ys = np.arange(h, dtype=np.float32) / cell
    xs = np.arange(w, dtype=np.float32) / cell
    y0 = ys.astype(np.int32); x0 = xs.astype(np.int32)
    fy = (ys - y0)[:, None]; fx = (xs - x0)[None, :]
    fy = fy * fy * (3 - 2 * fy); fx = fx * fx * (3 - 2 * fx)
    a = g[y0][:, x0]; b = g[y0][:, x0 + 1]; c = g[y0 + 1][:, x0]; d = g[y0 + 1][:, x0 + 1]
    return (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy


def make_frame(h: int, w: int, seed: int, n_rects: int = 200) -> np.ndarray:
    """One u8 [h,w] frame."""
    rng = np.random.Generator(np.random.PCG64(seed))
    img = np.zeros((h, w), np.float32)
    amp = 1.0
    for cell in (64, 32, 16, 8, 4):
        img += amp * _value_noise(rng, h, w, cell)
        amp *= 0.55
    img = img / img.max()
    for _ in range(n_rects):
        rh = int(rng.integers(4, max(5, h // 6))); rw = int(rng.integers(4, max(5, w // 10)))
        y = int(rng.integers(0, max(1, h - rh))); x = int(rng.integers(0, max(1, w - rw)))
        img[y:y + rh, x:x + rw] = 0.35 * img[y:y + rh, x:x + rw] + 0.65 * float(rng.random())
    # 3x3 binomial blur (sigma ~ 0.85)
    p = np.pad(img, 1, mode="edge")
    img = (p[:-2, :-2] + 2 * p[:-2, 1:-1] + p[:-2, 2:] + 2 * p[1:-1, :-2] + 4 * p[1:-1, 1:-1]
           + 2 * p[1:-1, 2:] + p[2:, :-2] + 2 * p[2:, 1:-1] + p[2:, 2:]) / 16.0
    return np.clip(img * 255.0 + 0.5, 0, 255).astype(np.uint8)


def make_stereo_pair(h: int, w: int, seed: int, band: int = 47):
    """(left, right) u8 [h,w]; right row-band r is left shifted left by d_r in [4, 60] px."""
    left = make_frame(h, w, seed)
    rng = np.random.Generator(np.random.PCG64(seed + 0x9E3779B1))
    right = np.empty_like(left)
    for y0 in range(0, h, band):
        d = int(rng.integers(4, 61))
        blk = left[y0:y0 + band]
        shifted = np.empty_like(blk)
        shifted[:, : w - d] = blk[:, d:]
        shifted[:, w - d:] = blk[:, -1:]
        right[y0:y0 + band] = shifted
    return left, right
