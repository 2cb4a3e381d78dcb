#!/usr/bin/env python3
"""Derive the float32 polynomial coefficients of the rt355 math specification.

The reference leaves sin/cos/atan2/asin/pow/tan to the GPU driver (SURVEY.md §8c,
App. B D12).  rt355 fixes them as explicit float32 polynomial kernels so the CPU
oracle and the HIP kernel execute the same IEEE op sequence.  This script fits
the polynomials (weighted least squares on Chebyshev nodes, float64), rounds
the coefficients to float32 and prints them as C hex-float literals.  It is a
one-off derivation tool; the printed literals are pasted into
oracle/rt_oracle_math.h and rsraytracingv3_amd/csrc/rt_math.h and pinned by
tests/test_oracle_math.py (max-ulp error against float64 libm).
"""
import numpy as np


def cheb_nodes(a, b, n):
    k = np.arange(n)
    x = np.cos(np.pi * (2 * k + 1) / (2 * n))
    return 0.5 * (a + b) + 0.5 * (b - a) * x


def fit(fn, a, b, deg, weight=None, n=4000):
    """Fit sum c_k t^k ~ fn(t) on [a,b] minimising weighted error; iterate Lawson
    re-weighting a few times to approach the minimax solution."""
    t = cheb_nodes(a, b, n)
    y = fn(t)
    w = np.ones_like(t) if weight is None else weight(t)
    V = np.vander(t, deg + 1, increasing=True)
    lw = np.ones_like(t)
    c = None
    for _ in range(60):
        ww = w * lw
        c, *_ = np.linalg.lstsq(V * ww[:, None], y * ww, rcond=None)
        err = np.abs((V @ c - y) * w)
        lw = lw * (0.5 + err / err.max())
        lw /= lw.max()
    return c.astype(np.float32), float(np.abs((V @ c - y) * w).max())


def show(name, c):
    print(f"// {name}")
    for i, v in enumerate(c):
        print(f"  c{i} = {float(v).hex()}f  /* {float(v):.9e} */")


if __name__ == "__main__":
    np.seterr(all="ignore")
    q = (np.pi / 4) ** 2
    # sin(x) = x + x*z*P(z), z = x*x
    def sin_p(z):
        x = np.sqrt(np.maximum(z, 1e-300))
        return np.where(z < 1e-12, -1 / 6.0, (np.sin(x) / x - 1) / z)
    c, e = fit(sin_p, 0, q * 1.02, 2, weight=lambda z: z)
    show(f"SIN P(z) deg2, rel err {e:.3e}", c)
    # cos(x) = 1 - z/2 + z*z*Q(z)
    def cos_q(z):
        x = np.sqrt(np.maximum(z, 1e-300))
        return np.where(z < 1e-6, 1 / 24.0, (np.cos(x) - 1 + z / 2) / (z * z))
    c, e = fit(cos_q, 0, q * 1.02, 2, weight=lambda z: z * z)
    show(f"COS Q(z) deg2, abs err {e:.3e}", c)
    # atan(a) = a + a*s*R(s), s = a*a, a in [0,1]
    def atan_r(s):
        a = np.sqrt(np.maximum(s, 1e-300))
        return np.where(s < 1e-10, -1 / 3.0, (np.arctan(a) / a - 1) / s)
    for deg in (6, 7, 8):
        c, e = fit(atan_r, 0, 1, deg, weight=lambda s: s)
        show(f"ATAN R(s) deg{deg}, rel err {e:.3e}", c)
    # asin(x) = x + x*z*P(z), z = x*x in [0, 0.25]
    def asin_p(z):
        x = np.sqrt(np.maximum(z, 1e-300))
        return np.where(z < 1e-10, 1 / 6.0, (np.arcsin(x) / x - 1) / z)
    for deg in (4, 5):
        c, e = fit(asin_p, 0, 0.25, deg, weight=lambda z: z)
        show(f"ASIN P(z) deg{deg}, rel err {e:.3e}", c)
    # log2(1+f) = f * L(f), f in [sqrt(.5)-1, sqrt(2)-1]
    def log2_l(f):
        return np.where(np.abs(f) < 1e-12, 1 / np.log(2.0), np.log2(1 + f) / f)
    for deg in (8, 9):
        c, e = fit(log2_l, np.sqrt(0.5) - 1, np.sqrt(2.0) - 1, deg)
        show(f"LOG2 L(f) deg{deg}, rel err {e:.3e}", c)
    # exp2(f) = 1 + f*E(f), f in [-0.5, 0.5]
    def exp2_e(f):
        return np.where(np.abs(f) < 1e-12, np.log(2.0), (np.exp2(f) - 1) / f)
    for deg in (5, 6):
        c, e = fit(exp2_e, -0.5, 0.5, deg, weight=lambda f: np.abs(f) + 1e-3)
        show(f"EXP2 E(f) deg{deg}, abs err {e:.3e}", c)
