// rt_math.h — float32 arithmetic of the rt355 render path (host + gfx950 device).
//
// The reference shader (src/shaders/*.hlsl) leaves sin/cos/tan/atan2/asin/pow, normalize,
// FMA contraction and constant-division to the GPU driver.  rt355 fixes each of them as a
// sequence of IEEE-754 binary32 operations (DESIGN.md "Arithmetic specification") so that
// results do not depend on the compiler, the device or the launch geometry, and so that
// they can be checked bit-for-bit against the independent CPU oracle.
//
// Compile rule: -ffp-contract=off for host AND device passes.  Every fused multiply-add is
// an explicit fmaf() (v_fma_f32 on gfx950); `/` and sqrtf() are the correctly rounded forms
// (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt); f32 denormals are kept (gfx950
// default float mode).  The same functions run on the host for the per-frame camera values,
// the per-object material constants and the sRGB decode table.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define RT_HD __host__ __device__ __forceinline__
#else
#define RT_HD inline
#endif

namespace rt355 {

constexpr float kPi = 3.141592654f;          // utils.hlsl:1
constexpr float kTwoPi = 6.2831855f;         // 2.*PI (random.hlsl:35), an exact doubling
constexpr float kEpsilon = 0.000001f;        // utils.hlsl:2
constexpr float kInvTwoPi = 0.15915494f;     // spec D16: x / c  ==  x * (1/c rounded)
constexpr float kInvPi = 0.31830987f;
constexpr float kInv180 = 0.0055555557f;
constexpr float kHalfPi = 1.57079637f;
constexpr float kFltMin = 1.17549435e-38f;

struct f3 {
  float x, y, z;
};

RT_HD f3 mk3(float x, float y, float z) { return f3{x, y, z}; }
RT_HD f3 operator+(f3 a, f3 b) { return f3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_HD f3 operator-(f3 a, f3 b) { return f3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_HD f3 operator*(f3 a, f3 b) { return f3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_HD f3 operator*(f3 a, float s) { return f3{a.x * s, a.y * s, a.z * s}; }
RT_HD f3 operator-(f3 a) { return f3{-a.x, -a.y, -a.z}; }
// a*s + b with one rounding per component
RT_HD f3 fma3(f3 a, float s, f3 b) { return f3{fmaf(a.x, s, b.x), fmaf(a.y, s, b.y), fmaf(a.z, s, b.z)}; }
// a*b + c component-wise, one rounding per component
RT_HD f3 fma3(f3 a, f3 b, f3 c) { return f3{fmaf(a.x, b.x, c.x), fmaf(a.y, b.y, c.y), fmaf(a.z, b.z, c.z)}; }
// x, then y and z fused in, in that order
RT_HD float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
RT_HD f3 cross3(f3 a, f3 b) {
  return f3{fmaf(a.y, b.z, -(a.z * b.y)), fmaf(a.z, b.x, -(a.x * b.z)), fmaf(a.x, b.y, -(a.y * b.x))};
}
// The correctly rounded square root of the shading code.  A device translation unit may define RT_SQRTF as its own
// correctly rounded form before including this header (rt_kernels.hip: sqrt_rn, 11 instructions instead of hipcc's 16).
#ifndef RT_SQRTF
#define RT_SQRTF(x) sqrtf(x)
#endif
RT_HD float length3(f3 a) { return RT_SQRTF(dot3(a, a)); }
// 1 / s for s a correctly rounded square root (0, +inf, NaN, or 2^-75 .. 2^64): IEEE division; a device translation
// unit may define RT_RCP_OF_ROOT as its own correctly rounded form (rt_kernels.hip: rcp_of_root, 4 instructions for 11).
#ifndef RT_RCP_OF_ROOT
#define RT_RCP_OF_ROOT(s) (1.0f / (s))
#endif
RT_HD f3 normalize3(f3 a) {
  const float inv = RT_RCP_OF_ROOT(RT_SQRTF(dot3(a, a)));
  return a * inv;
}
RT_HD float saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }

RT_HD uint32_t f2u(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __float_as_uint(x);
#else
  uint32_t u;
  std::memcpy(&u, &x, 4);
  return u;
#endif
}
RT_HD float u2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __uint_as_float(u);
#else
  float x;
  std::memcpy(&x, &u, 4);
  return x;
#endif
}

// sin(x), cos(x): quadrant k = floor(x*(2/pi) + 1/2); y = x - k*pi/2 as two fma steps
// (pi/2 = hi + lo); polynomial kernels on |y| <= pi/4; quadrant swap/negate.
RT_HD void sincos_spec(float x, float &s, float &c) {
  const float kf = floorf(fmaf(x, 0x1.45f306p-1f, 0.5f));
  float y = fmaf(-kf, 0x1.921fb6p+0f, x);
  y = fmaf(-kf, -0x1.777a5cp-25f, y);
  const float z = y * y;
  float ps = fmaf(z, -0x1.992c30p-13f, 0x1.1106fcp-7f);
  ps = fmaf(z, ps, -0x1.555544p-3f);
  ps = fmaf(y * z, ps, y);
  float pc = fmaf(z, 0x1.99e360p-16f, -0x1.6c0c22p-10f);
  pc = fmaf(z, pc, 0x1.55554ap-5f);
  pc = fmaf(z * z, pc, fmaf(-0.5f, z, 1.0f));
  const int k = static_cast<int>(kf) & 3;
  const float s0 = (k & 1) ? pc : ps;
  const float c0 = (k & 1) ? ps : pc;
  s = (k & 2) ? -s0 : s0;
  c = ((k + 1) & 2) ? -c0 : c0;
}

RT_HD float tan_spec(float x) {
  float s, c;
  sincos_spec(x, s, c);
  return s / c;
}

// atan2(y, x) from a = min(|x|,|y|) / max(|x|,|y|), atan(a) = a + a*s*R(s), s = a*a.
RT_HD float atan2_spec(float y, float x) {
  const float ax = fabsf(x), ay = fabsf(y);
  const float hi = fmaxf(ax, ay), lo = fminf(ax, ay);
  const float a = (hi == 0.0f) ? 0.0f : lo / hi;
  const float s = a * a;
  float r = fmaf(s, 0x1.85e1f2p-9f, -0x1.0faa28p-6f);
  r = fmaf(s, r, 0x1.64ca1ap-5f);
  r = fmaf(s, r, -0x1.368000p-4f);
  r = fmaf(s, r, 0x1.b568c0p-4f);
  r = fmaf(s, r, -0x1.231b62p-3f);
  r = fmaf(s, r, 0x1.997adcp-3f);
  r = fmaf(s, r, -0x1.5554e6p-2f);
  r = fmaf(a * s, r, a);
  r = (ay > ax) ? kHalfPi - r : r;
  r = (x < 0.0f) ? kPi - r : r;
  return (y < 0.0f) ? -r : r;
}

// asin(x), argument clamped to [-1, 1] (App. B D8).
RT_HD float asin_spec(float x) {
  const float ax = fminf(fabsf(x), 1.0f);
  const bool big = ax > 0.5f;
  const float z = big ? fmaf(-0.5f, ax, 0.5f) : ax * ax;
  const float t = big ? RT_SQRTF(z) : ax;
  float p = fmaf(z, 0x1.59dabcp-5f, 0x1.8ba266p-6f);
  p = fmaf(z, p, 0x1.748c64p-5f);
  p = fmaf(z, p, 0x1.33014ep-4f);
  p = fmaf(z, p, 0x1.5555cap-3f);
  float r = fmaf(t * z, p, t);
  r = big ? fmaf(-2.0f, r, kHalfPi) : r;
  return (x < 0.0f) ? -r : r;
}

// log2(x) for x >= FLT_MIN; -150 otherwise.
RT_HD float log2_spec(float x) {
  if (!(x >= kFltMin)) return -150.0f;
  const uint32_t ix = f2u(x);
  int e = static_cast<int>(ix >> 23) - 127;
  float m = u2f((ix & 0x007fffffu) | 0x3f800000u);
  if (m > 1.41421354f) {
    m = m * 0.5f;
    e += 1;
  }
  const float f = m - 1.0f;
  float p = fmaf(f, -0x1.cebe00p-4f, 0x1.80ab18p-3f);
  p = fmaf(f, p, -0x1.865ffcp-3f);
  p = fmaf(f, p, 0x1.a265fcp-3f);
  p = fmaf(f, p, -0x1.eac694p-3f);
  p = fmaf(f, p, 0x1.2782e6p-2f);
  p = fmaf(f, p, -0x1.715a68p-2f);
  p = fmaf(f, p, 0x1.ec7080p-2f);
  p = fmaf(f, p, -0x1.715470p-1f);
  p = fmaf(f, p, 0x1.715476p+0f);
  return fmaf(f, p, static_cast<float>(e));
}

// exp2(z): k = floor(z + 1/2), f = z - k, 2^f = 1 + f*E(f), then add k to the exponent field.
RT_HD float exp2_spec(float z) {
  const float kf = floorf(z + 0.5f);
  const float f = z - kf;
  float p = fmaf(f, 0x1.448f04p-13f, 0x1.5f477cp-10f);
  p = fmaf(f, p, 0x1.3b29acp-7f);
  p = fmaf(f, p, 0x1.c6aed0p-5f);
  p = fmaf(f, p, 0x1.ebfbe0p-3f);
  p = fmaf(f, p, 0x1.62e430p-1f);
  p = fmaf(f, p, 1.0f);
  if (!(kf >= -125.0f)) return 0.0f;
  if (kf > 127.0f) return INFINITY;
  return u2f(f2u(p) + (static_cast<uint32_t>(static_cast<int32_t>(kf)) << 23));
}

// pow(x, y), y > 0: exp2(y * log2(x)); 0 for x < FLT_MIN.
RT_HD float pow_spec(float x, float y) {
  if (!(x >= kFltMin)) return 0.0f;
  return exp2_spec(y * log2_spec(x));
}

// Rgba8UnormSrgb store / load (render_target.rs:23; App. B D13).
RT_HD uint32_t srgb8_encode_spec(float c) {
  c = saturate(c);  // NaN -> 0
  const float s = (c <= 0.0031308f) ? 12.92f * c : fmaf(1.055f, pow_spec(c, 0.41666666f), -0.055f);
  return static_cast<uint32_t>(floorf(fmaf(255.0f, s, 0.5f)));
}
RT_HD float srgb8_decode_spec(uint32_t b) {
  const float s = static_cast<float>(b) / 255.0f;
  return (s <= 0.04045f) ? s / 12.92f : pow_spec((s + 0.055f) / 1.055f, 2.4f);
}

// Nearest, clamp-to-edge texel index (connection.rs:164-174): clamp(floor(u*size), 0, size-1).
RT_HD uint32_t texel_index(float u, uint32_t size) {
  float t = floorf(u * static_cast<float>(size));
  t = fminf(fmaxf(t, 0.0f), static_cast<float>(size - 1u));
  return static_cast<uint32_t>(t);
}

// The same with the two conversions done by the caller (size_f = float(size), last_f = float(size - 1)): on the device
// they are wave-uniform kernel arguments, read as scalar operands instead of being converted — and kept — in vector
// registers.
RT_HD uint32_t texel_index(float u, float size_f, float last_f) {
  float t = floorf(u * size_f);
#if defined(__HIP_DEVICE_COMPILE__)
  // the clamp as one v_med3_f32: the median of (t, 0, last) is fminf(fmaxf(t, 0), last) for every t including the
  // infinities, and for a NaN the instruction returns the minimum of the other two operands, 0, as fmaxf(NaN, 0) does
  // (tests/test_parity_gpu.py::test_device_math_bitwise)
  t = __builtin_amdgcn_fmed3f(t, 0.0f, last_f);
#else
  t = fminf(fmaxf(t, 0.0f), last_f);
#endif
  return static_cast<uint32_t>(t);
}

}  // namespace rt355
