/* rt_oracle_math.h — float32 math of the CPU oracle (TEST INFRASTRUCTURE, not product).
 *
 * The reference leaves its transcendental functions to the GPU driver behind
 * wgpu (SURVEY.md §8c "transcendental precision", App. B D12): every intrinsic
 * used by src/shaders/{frag,ray,random,utils}.hlsl is implementation-defined to
 * a few ulp.  The oracle fixes each one as an explicit sequence of IEEE-754
 * binary32 operations (+, -, *, fma, /, sqrt, floor, compare, integer bit
 * moves) so that a second, independent implementation of the same written
 * specification (rsraytracingv3_amd/csrc/rt_math.h, compiled for gfx950) can be
 * compared BIT FOR BIT.  Coefficients come from tools/fit_math.py; accuracy
 * against float64 libm is pinned by tests/test_oracle_math.py.
 *
 * Build rule: -ffp-contract=off (no implicit fusing); every fused multiply-add
 * is written fmaf() explicitly.  This file is plain C99.
 */
#ifndef RT_ORACLE_MATH_H
#define RT_ORACLE_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

/* utils.hlsl:1-2 — literals as the HLSL front-end rounds them to binary32. */
#define RTO_PI 3.141592654f      /* -> 0x1.921fb6p+1 */
#define RTO_TWO_PI 6.2831855f    /* 2.*PI, exact doubling of RTO_PI */
#define RTO_EPSILON 0.000001f
/* Spec decision D16: division by a compile-time constant is a
 * multiplication by its correctly rounded binary32 reciprocal. */
#define RTO_INV_TWO_PI 0.15915494f /* 1/(2*pi) */
#define RTO_INV_PI 0.31830987f     /* 1/pi     */
#define RTO_PIO2 1.57079637f       /* 0x1.921fb6p+0 */

static inline uint32_t rto_bits(float x) { uint32_t u; memcpy(&u, &x, 4); return u; }
static inline float rto_float(uint32_t u) { float x; memcpy(&x, &u, 4); return x; }

/* sin and cos of one argument.  Quadrant reduction k = floor(x*2/pi + 1/2),
 * y = x - k*pi/2 in two fma steps (Cody-Waite, pi/2 = HI + LO), then the
 * minimax kernels on |y| <= pi/4. */
static inline void rto_sincos(float x, float *s, float *c) {
  float kf = floorf(fmaf(x, 0x1.45f306p-1f /* 2/pi */, 0.5f));
  float y = fmaf(-kf, 0x1.921fb6p+0f /* pi/2 hi */, x);
  y = fmaf(-kf, -0x1.777a5cp-25f /* pi/2 lo */, y);
  float z = y * y;
  float p = fmaf(z, -0x1.992c30p-13f, 0x1.1106fcp-7f);
  p = fmaf(z, p, -0x1.555544p-3f);
  float ps = fmaf(y * z, p, y);
  float q = fmaf(z, 0x1.99e360p-16f, -0x1.6c0c22p-10f);
  q = fmaf(z, q, 0x1.55554ap-5f);
  float pc = fmaf(z * z, q, fmaf(-0.5f, z, 1.0f));
  int k = (int)kf;
  switch (k & 3) {
    case 0: *s = ps; *c = pc; break;
    case 1: *s = pc; *c = -ps; break;
    case 2: *s = -ps; *c = -pc; break;
    default: *s = -pc; *c = ps; break;
  }
}

static inline float rto_tan(float x) {
  float s, c;
  rto_sincos(x, &s, &c);
  return s / c;
}

/* atan2(y, x): a = min/max in [0,1], atan(a) = a + a*s*R(s), octant fix-up. */
static inline float rto_atan2(float y, float x) {
  float ax = fabsf(x), ay = fabsf(y);
  float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
  float a = (mx == 0.0f) ? 0.0f : mn / mx;
  float s = a * a;
  float r = fmaf(s, 0x1.85e1f2p-9f, -0x1.0faa28p-6f);
  r = fmaf(s, r, 0x1.64ca1ap-5f);
  r = fmaf(s, r, -0x1.368000p-4f);
  r = fmaf(s, r, 0x1.b568c0p-4f);
  r = fmaf(s, r, -0x1.231b62p-3f);
  r = fmaf(s, r, 0x1.997adcp-3f);
  r = fmaf(s, r, -0x1.5554e6p-2f);
  r = fmaf(a * s, r, a);
  if (ay > ax) r = RTO_PIO2 - r;
  if (x < 0.0f) r = RTO_PI - r;
  return (y < 0.0f) ? -r : r;
}

/* asin(x) with the argument clamped to [-1,1] first (App. B D8). */
static inline float rto_asin(float x) {
  float ax = fminf(fabsf(x), 1.0f);
  float t, z;
  int hi = ax > 0.5f;
  if (hi) {
    z = fmaf(-0.5f, ax, 0.5f);
    t = sqrtf(z);
  } else {
    z = ax * ax;
    t = ax;
  }
  float p = fmaf(z, 0x1.59dabcp-5f, 0x1.8ba266p-6f);
  p = fmaf(z, p, 0x1.748c64p-5f);
  p = fmaf(z, p, 0x1.33014ep-4f);
  p = fmaf(z, p, 0x1.5555cap-3f);
  float r = fmaf(t * z, p, t);
  if (hi) r = fmaf(-2.0f, r, RTO_PIO2);
  return (x < 0.0f) ? -r : r;
}

/* log2(x) for x >= FLT_MIN; anything else (0, negatives, subnormals, NaN)
 * returns -150 (callers treat that as "pow result 0"). */
static inline float rto_log2(float x) {
  if (!(x >= 1.17549435e-38f)) return -150.0f;
  uint32_t ix = rto_bits(x);
  int e = (int)(ix >> 23) - 127;
  float m = rto_float((ix & 0x007fffffu) | 0x3f800000u);
  if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
  float f = m - 1.0f;
  float p = fmaf(f, -0x1.cebe00p-4f, 0x1.80ab18p-3f);
  p = fmaf(f, p, -0x1.865ffcp-3f);
  p = fmaf(f, p, 0x1.a265fcp-3f);
  p = fmaf(f, p, -0x1.eac694p-3f);
  p = fmaf(f, p, 0x1.2782e6p-2f);
  p = fmaf(f, p, -0x1.715a68p-2f);
  p = fmaf(f, p, 0x1.ec7080p-2f);
  p = fmaf(f, p, -0x1.715470p-1f);
  p = fmaf(f, p, 0x1.715476p+0f);
  return fmaf(f, p, (float)e);
}

/* exp2(z): k = floor(z + 1/2), f = z - k, 2^f = 1 + f*E(f), exponent add. */
static inline float rto_exp2(float z) {
  float kf = floorf(z + 0.5f);
  float f = z - kf;
  float p = fmaf(f, 0x1.448f04p-13f, 0x1.5f477cp-10f);
  p = fmaf(f, p, 0x1.3b29acp-7f);
  p = fmaf(f, p, 0x1.c6aed0p-5f);
  p = fmaf(f, p, 0x1.ebfbe0p-3f);
  p = fmaf(f, p, 0x1.62e430p-1f);
  p = fmaf(f, p, 1.0f);
  if (!(kf >= -125.0f)) return 0.0f; /* also NaN */
  if (kf > 127.0f) return INFINITY;
  return rto_float(rto_bits(p) + ((uint32_t)(int32_t)kf << 23));
}

/* pow(x, y) for y > 0, as the Vulkan/SPIR-V definition exp2(y*log2(x));
 * x < FLT_MIN (incl. 0 and negatives) gives 0. */
static inline float rto_pow(float x, float y) {
  if (!(x >= 1.17549435e-38f)) return 0.0f;
  return rto_exp2(y * rto_log2(x));
}

#endif /* RT_ORACLE_MATH_H */
