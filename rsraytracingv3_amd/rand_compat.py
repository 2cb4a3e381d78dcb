"""Bit-compatible re-implementation of the random streams the reference's scene generator draws
from: ``rand 0.8.5``'s ``StdRng`` (= ``rand_chacha 0.3.1`` ChaCha12, 64-bit counter,
four blocks per refill), ``SeedableRng::seed_from_u64`` (PCG32 seed expansion, rand_core 0.6), the
``Standard`` f32 distribution, ``gen_range`` for ``usize`` (widening multiply + zone rejection) and
for an inclusive f32 range, and ``rand_distr 0.4.3``'s ``UnitDisc``.

None of those crates is in ``/root/reference`` (they are Cargo.lock dependencies), so the algorithms
are restated from their published sources; the pin is the reference's own screenshot, whose object
panel shows Sphere 0 of the default seed-42 scene (``screenshots/screenshot.png``): position
(2.7, 3.7, -25.1), radius 3.7, emission yellow, strength 5.34, metallic 0.617, roughness 0.348 —
``tests/test_rand_compat.py``.
"""
from __future__ import annotations

import struct
from typing import List, Tuple

import numpy as np

_M32 = 0xFFFFFFFF
_M64 = 0xFFFFFFFFFFFFFFFF


def fma_f32(a, b, c) -> np.float32:
    """``f32::mul_add`` (scene.rs:105-107, connection.rs:392-399): a*b + c rounded ONCE to binary32,
    nearest-even.  Evaluated exactly in rational arithmetic, so neither the product nor a float64
    intermediate rounds first (float64 would round twice).  Normal-range results only."""
    from fractions import Fraction
    x = Fraction(float(np.float32(a))) * Fraction(float(np.float32(b))) + Fraction(float(np.float32(c)))
    if x == 0:
        return np.float32(0.0)
    n = abs(x)
    e = n.numerator.bit_length() - n.denominator.bit_length()  # 2^(e-1) < n < 2^(e+1)
    if Fraction(2) ** e > n:
        e -= 1                                                  # now 2^e <= n < 2^(e+1)
    q = n / Fraction(2) ** (e - 23)                             # in [2^23, 2^24)
    m = q.numerator // q.denominator
    rem = q - m
    if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and (m & 1)):
        m += 1
    v = float(m) * 2.0 ** (e - 23)                              # exact: m <= 2^24
    out = np.float32(-v if x < 0 else v)
    assert -126 <= e <= 127 and float(out) == (-v if x < 0 else v)
    return out


def _rotl32(x: int, n: int) -> int:
    return ((x << n) | (x >> (32 - n))) & _M32


def _chacha_block(key: List[int], counter: int, stream: int, rounds: int) -> List[int]:
    init = [0x61707865, 0x3320646E, 0x79622D32, 0x6B206574] + key + [
        counter & _M32, (counter >> 32) & _M32, stream & _M32, (stream >> 32) & _M32]
    x = list(init)

    def qr(a, b, c, d):
        x[a] = (x[a] + x[b]) & _M32; x[d] = _rotl32(x[d] ^ x[a], 16)
        x[c] = (x[c] + x[d]) & _M32; x[b] = _rotl32(x[b] ^ x[c], 12)
        x[a] = (x[a] + x[b]) & _M32; x[d] = _rotl32(x[d] ^ x[a], 8)
        x[c] = (x[c] + x[d]) & _M32; x[b] = _rotl32(x[b] ^ x[c], 7)

    for _ in range(rounds // 2):
        qr(0, 4, 8, 12); qr(1, 5, 9, 13); qr(2, 6, 10, 14); qr(3, 7, 11, 15)
        qr(0, 5, 10, 15); qr(1, 6, 11, 12); qr(2, 7, 8, 13); qr(3, 4, 9, 14)
    return [(x[i] + init[i]) & _M32 for i in range(16)]


class StdRng:
    """``rand::rngs::StdRng`` of rand 0.8: ChaCha12 behind a 64-word ``BlockRng``."""

    ROUNDS = 12
    BUF = 64  # four 16-word blocks per refill (rand_chacha's BUFSZ)

    def __init__(self, seed32: bytes):
        assert len(seed32) == 32
        self.key = list(struct.unpack("<8I", seed32))
        self.counter = 0
        self.results: List[int] = [0] * self.BUF
        self.index = self.BUF  # empty

    @staticmethod
    def seed_from_u64(state: int) -> "StdRng":
        """rand_core 0.6 ``SeedableRng::seed_from_u64``: PCG32 output per 4-byte chunk of the seed."""
        mul, inc = 6364136223846793005, 11634580027462260723
        seed = b""
        for _ in range(8):
            state = (state * mul + inc) & _M64
            xorshifted = (((state >> 18) ^ state) >> 27) & _M32
            rot = state >> 59
            x = ((xorshifted >> rot) | (xorshifted << ((32 - rot) & 31))) & _M32
            seed += struct.pack("<I", x)
        return StdRng(seed)

    def _generate(self):
        out: List[int] = []
        for b in range(4):
            out += _chacha_block(self.key, self.counter + b, 0, self.ROUNDS)
        self.counter += 4
        self.results = out

    def next_u32(self) -> int:
        if self.index >= self.BUF:
            self._generate()
            self.index = 0
        v = self.results[self.index]
        self.index += 1
        return v

    def next_u64(self) -> int:
        """``BlockRng::next_u64``: two consecutive words, low first, with the buffer-edge cases."""
        if self.index < self.BUF - 1:
            lo, hi = self.results[self.index], self.results[self.index + 1]
            self.index += 2
        elif self.index >= self.BUF:
            self._generate()
            lo, hi = self.results[0], self.results[1]
            self.index = 2
        else:
            lo = self.results[self.BUF - 1]
            self._generate()
            hi = self.results[0]
            self.index = 1
        return (hi << 32) | lo

    # -- distributions ---------------------------------------------------------------------------
    def gen_f32(self) -> np.float32:
        """``Standard`` for f32: 24 high bits times 2^-24."""
        return np.float32(self.next_u32() >> 8) * np.float32(1.0 / (1 << 24))

    def gen_range_usize(self, low: int, high: int) -> int:
        """``gen_range(low..high)`` for usize on a 64-bit target (UniformInt::sample_single)."""
        rng_range = (high - 1 - low + 1) & _M64
        lz = 64 - rng_range.bit_length()
        zone = (((rng_range << lz) & _M64) - 1) & _M64
        while True:
            v = self.next_u64()
            prod = v * rng_range
            hi, lo = prod >> 64, prod & _M64
            if lo <= zone:
                return low + hi

    def _value0_1(self) -> np.float32:
        bits = (self.next_u32() >> 9) | 0x3F800000  # into_float_with_exponent(0): [1, 2)
        return np.float32(struct.unpack("<f", struct.pack("<I", bits))[0]) - np.float32(1.0)

    def uniform_f32(self, low: float, high: float) -> np.float32:
        """``Uniform::new(low, high).sample`` for f32 (half-open)."""
        low, high = np.float32(low), np.float32(high)
        max_rand = np.float32(1.0) - np.float32(2.0 ** -23)
        scale = np.float32(high - low)
        while np.float32(scale * max_rand + low) >= high:
            scale = np.frombuffer(struct.pack("<I", struct.unpack("<I", struct.pack("<f", scale))[0] - 1), np.float32)[0]
        return np.float32(self._value0_1() * scale + low)

    def gen_range_f32_inclusive(self, low: float, high: float) -> np.float32:
        """``gen_range(low..=high)`` for f32 (UniformFloat::new_inclusive + sample)."""
        low, high = np.float32(low), np.float32(high)
        max_rand = np.float32(1.0) - np.float32(2.0 ** -23)
        scale = np.float32(np.float32(high - low) / max_rand)
        while np.float32(scale * max_rand + low) > high:
            scale = np.frombuffer(struct.pack("<I", struct.unpack("<I", struct.pack("<f", scale))[0] - 1), np.float32)[0]
        return np.float32(self._value0_1() * scale + low)

    def unit_disc(self) -> Tuple[np.float32, np.float32]:
        """``rand_distr::UnitDisc``: rejection from the square [-1, 1)^2."""
        while True:
            x1 = self.uniform_f32(-1.0, 1.0)
            x2 = self.uniform_f32(-1.0, 1.0)
            if np.float32(np.float32(x1 * x1) + np.float32(x2 * x2)) <= np.float32(1.0):
                return x1, x2
