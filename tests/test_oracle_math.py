"""Pins the oracle's fixed float32 math (oracle/rt_oracle_math.h) against float64 libm:
each function stays within a few ulp of the true value over the ranges the shader uses, so
"sin" really is sin.  (The reference leaves these to the GPU driver — SURVEY.md §8c; Vulkan
only promises e.g. 2^-11 absolute for sin/cos and exp2(y*log2 x) quality for pow.)"""
import numpy as np
import pytest

import oracle


def ulp_error(got, ref64):
    ulp = np.spacing(np.abs(ref64).astype(np.float32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / ulp


RS = np.random.RandomState(0)
N = 400_000


@pytest.mark.parametrize("name,fn,lo,hi,bound", [
    ("sin", np.sin, -7.0, 7.0, 2.0),
    ("cos", np.cos, -7.0, 7.0, 2.0),
    ("tan", np.tan, 0.0, 0.79, 3.5),      # fov/2 of a fov in (0, 90] degrees
    ("asin", np.arcsin, -1.0, 1.0, 3.0),
    ("exp2", np.exp2, -30.0, 30.0, 1.5),
])
def test_unary_within_ulps(name, fn, lo, hi, bound):
    x = RS.uniform(lo, hi, N).astype(np.float32)
    err = ulp_error(oracle.math_unary(name, x), fn(x.astype(np.float64)))
    assert err.max() <= bound, "%s: %.2f ulp" % (name, err.max())


def test_log2():
    x = np.exp(RS.uniform(-20, 20, N)).astype(np.float32)
    got = oracle.math_unary("log2", x)
    ref = np.log2(x.astype(np.float64))
    assert np.abs(got - ref).max() <= 2e-6
    assert ulp_error(got, ref).max() <= 2.5
    # exact at powers of two; the documented sentinel for non-positive / subnormal input
    p2 = np.array([0.25, 0.5, 1.0, 2.0, 1024.0], np.float32)
    assert np.array_equal(oracle.math_unary("log2", p2), np.array([-2, -1, 0, 1, 10], np.float32))
    bad = np.array([0.0, -1.0, 1e-40, np.nan], np.float32)
    assert np.array_equal(oracle.math_unary("log2", bad), np.full(4, -150, np.float32))


def test_atan2_all_quadrants():
    a = RS.normal(size=N).astype(np.float32)
    b = RS.normal(size=N).astype(np.float32)
    err = ulp_error(oracle.math_binary("atan2", a, b), np.arctan2(a.astype(np.float64), b.astype(np.float64)))
    assert err.max() <= 2.5
    ax = np.array([0, 0, 1, -1, 0], np.float32)
    bx = np.array([1, -1, 0, 0, 0], np.float32)
    got = oracle.math_binary("atan2", ax, bx)
    assert got[0] == 0.0 and abs(got[1] - np.pi) < 1e-6 and abs(got[2] - np.pi / 2) < 1e-6
    assert abs(got[3] + np.pi / 2) < 1e-6 and got[4] == 0.0  # App. B D8: atan2(0,0) = 0


def test_pow_ranges_the_shader_uses():
    # pow(random(), 1/(alpha+1)): base in [0,1), exponent in (1/1001, 1/2]  (random.hlsl:33)
    a = RS.uniform(0, 1, N).astype(np.float32)
    b = RS.uniform(1 / 1001, 0.5, N).astype(np.float32)
    err = ulp_error(oracle.math_binary("pow", a, b), np.power(a.astype(np.float64), b.astype(np.float64)))
    assert err.max() <= 8.0
    # pow(1000., smoothness^2) (random.hlsl:31)
    s = RS.uniform(0, 1, N).astype(np.float32)
    k = np.full_like(s, 1000.0)
    err = ulp_error(oracle.math_binary("pow", k, s), np.power(1000.0, s.astype(np.float64)))
    assert err.max() <= 10.0
    # sRGB transfer exponents
    c = RS.uniform(0.003, 1, N).astype(np.float32)
    e = np.full_like(c, np.float32(0.41666666))
    err = ulp_error(oracle.math_binary("pow", c, e), np.power(c.astype(np.float64), float(np.float32(0.41666666))))
    assert err.max() <= 4.0
    z = oracle.math_binary("pow", np.array([0.0, -2.0, 1e-40], np.float32), np.array([0.5, 0.5, 0.5], np.float32))
    assert np.array_equal(z, np.zeros(3, np.float32))
    one = oracle.math_binary("pow", np.array([1.0], np.float32), np.array([2.4], np.float32))
    assert one[0] == 1.0


def test_asin_clamps_out_of_range_argument():
    x = np.array([1.0000001, -1.0000001, 3.0], np.float32)
    got = oracle.math_unary("asin", x)
    assert np.allclose(got, [np.pi / 2, -np.pi / 2, np.pi / 2], atol=2e-7)  # App. B D8


def test_energy_cutoff_constant_is_the_exact_root_threshold():
    """rt_kernels.hip replaces `length(energy) < EPSILON` (frag.hlsl:40-42) by `dot(energy, energy) < T`:
    T must be the smallest float whose correctly rounded square root reaches EPSILON."""
    import os
    import re
    eps = np.float32(0.000001)
    t = np.uint32(0x2b8cbccb).view(np.float32)
    below = np.nextafter(t, np.float32(0))
    assert np.sqrt(t, dtype=np.float32) >= eps and np.sqrt(below, dtype=np.float32) < eps
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rsraytracingv3_amd", "csrc",
                            "rt_kernels.hip")).read()
    assert re.search(r"dot3\(s\.energy, s\.energy\) < __uint_as_float\(0x2b8cbccbu\)", src)
