"""Seeded synthetic inputs for parity tests and the benchmark.

The reference draws its random texture from an unseeded ``fastrand`` stream on a
background thread (gpu/random_texture.rs:56-75) and its jitter from
``rand::random`` (gpu/connection.rs:392-399), and loads an EXR panorama that is
not shipped (``.MISSING_LARGE_BLOBS``).  Parity needs all three as explicit,
reproducible inputs, so they are generated here from one documented generator.
"""
from __future__ import annotations

import numpy as np

_MASK = (1 << 64) - 1


class SplitMix64:
    """SplitMix64 (Steele, Lea, Flood 2014).  ``f32`` keeps the top 24 bits, so every
    value is k·2⁻²⁴ in [0,1) — the same grid ``fastrand::Rng::f32`` /
    ``rand``'s ``Standard`` f32 produce (random_texture.rs:72)."""

    def __init__(self, seed: int):
        self.state = seed & _MASK

    def next_u64(self) -> int:
        self.state = (self.state + 0x9E3779B97F4A7C15) & _MASK
        z = self.state
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
        return z ^ (z >> 31)

    def f32(self) -> float:
        return (self.next_u64() >> 40) * (1.0 / (1 << 24))

    def below(self, n: int) -> int:
        return int((self.next_u64() >> 32) * n >> 32)


def _splitmix_block(seed: int, count: int) -> np.ndarray:
    """``count`` consecutive SplitMix64 outputs for ``seed``, vectorised."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, count + 1, dtype=np.uint64)
        z = np.uint64(seed & _MASK) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def random_texture(seed: int, width: int = 600, height: int = 600) -> np.ndarray:
    """R32Float noise texture, U[0,1) on the 2⁻²⁴ grid (random_texture.rs:21-22,67-75)."""
    bits = _splitmix_block(seed, width * height) >> np.uint64(40)
    return (bits.astype(np.float32) * np.float32(1.0 / (1 << 24))).reshape(height, width)


def jitter_sequence(seed: int, count: int) -> np.ndarray:
    """``count`` sub-pixel jitters: r·0.99 − 0.495 per axis (connection.rs:20,392-399)."""
    r = (_splitmix_block(seed ^ 0x6A09E667F3BCC909, 2 * count) >> np.uint64(40)).astype(np.float32)
    r = r * np.float32(1.0 / (1 << 24))
    return (r * np.float32(0.99) - np.float32(0.495)).reshape(count, 2)


def procedural_skybox(width: int, height: int, seed: int = 7) -> np.ndarray:
    """Equirectangular RGBA32F panorama standing in for ``table_mountain_1_8k.exr``
    (connection.rs:176-215): row 0 = top (zenith), alpha 1 (``into_rgba32f``).  Smooth
    sky gradient + darker ground, a bright sun lobe and a few hot texels, all
    positive HDR values in 0…~50."""
    v = (np.arange(height, dtype=np.float32) + 0.5) / height          # 0 top … 1 bottom
    u = (np.arange(width, dtype=np.float32) + 0.5) / width
    elev = (0.5 - v) * np.float32(np.pi)                               # +π/2 zenith
    azim = (u - 0.5) * np.float32(2 * np.pi)
    ce = np.cos(elev)[:, None]
    d = np.stack([ce * np.sin(azim)[None, :], np.broadcast_to(np.sin(elev)[:, None], (height, width)),
                  ce * np.cos(azim)[None, :]], axis=-1).astype(np.float32)
    up = np.clip(d[..., 1], 0, 1)
    down = np.clip(-d[..., 1], 0, 1)
    sky = np.stack([0.25 + 0.35 * (1 - up), 0.45 + 0.35 * (1 - up), 0.95 - 0.25 * (1 - up) * 0 + 0 * up], -1)
    ground = np.stack([0.28 - 0.1 * down, 0.24 - 0.08 * down, 0.2 - 0.06 * down], -1)
    img = np.where((d[..., 1] >= 0)[..., None], sky, ground).astype(np.float32)
    sun = np.array([0.45, 0.6, -0.66], dtype=np.float32)
    sun /= np.linalg.norm(sun)
    c = np.clip(d @ sun, 0, 1)
    img += (40.0 * c ** 600 + 1.5 * c ** 24)[..., None] * np.array([1.0, 0.93, 0.8], dtype=np.float32)
    # low-amplitude cloud bands so neighbouring texels differ
    img *= (1.0 + 0.08 * np.sin(9 * azim)[None, :, None] * np.cos(7 * elev)[:, None, None]).astype(np.float32)
    rng = SplitMix64(seed)
    for _ in range(8):  # hot texels
        j, i = rng.below(height // 2), rng.below(width)
        img[j, i, :3] += 20.0 + 25.0 * rng.f32()
    out = np.ones((height, width, 4), dtype=np.float32)
    out[..., :3] = np.maximum(img, 0.0)
    return np.ascontiguousarray(out)
