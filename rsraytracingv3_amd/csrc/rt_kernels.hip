// rt_kernels.hip — gfx950 kernels of the rt355 render path.
//
// One lane = one pixel of frag.hlsl:48-66 (fs_main).  A wave owns an 8x8 pixel tile so the
// 64 primary rays are coherent; a 256-thread workgroup stages the sphere records into LDS
// once and every ray segment scans them from there (wave-uniform address -> LDS broadcast,
// no bank conflicts).  Closest-hit selection follows ray.hlsl:97-107 exactly (App. B D10:
// strict '<' in ascending object order == lexicographic min of (distance, index)).
//
// The square root of the sphere test is only evaluated when some lane of the wave can still
// accept the sphere (ballot over "discriminant >= 0 and the sphere is in front"): rejected
// lanes of the reference never use the root, so skipping it is exact.
#include <hip/hip_runtime.h>

#include "rt_device.h"

namespace rt355 {

namespace {

constexpr int kBlock = 256;
constexpr int kTile = 8;  // 8x8 pixels per wave

struct Rng {
  float seed, px, py;  // random.hlsl:1-2
};

// random.hlsl:4-11.  fmod(seed, 600): seed stays in [0, 1200) for texture values in [0, 1),
// where one conditional subtraction is the exact remainder; anything else takes fmodf.
__device__ __forceinline__ float rng_next(const FrameParams &fp, Rng &rng) {
  const float sx = rng.px + rng.seed, sy = rng.py + rng.seed;
  const float cx = sx - floorf(sx), cy = sy - floorf(sy);
  const uint32_t i = texel_index(cx, fp.rnd_w), j = texel_index(cy, fp.rnd_h);
  const float x = fp.rnd[static_cast<size_t>(j) * fp.rnd_w + i];
  float s = fmaf(x, 532.3412f, rng.seed);
  if (s >= 0.0f && s < 1200.0f) {
    s = (s >= 600.0f) ? s - 600.0f : s;
  } else {
    s = fmodf(s, 600.0f);
  }
  rng.seed = s;
  return x;
}

// random.hlsl:13-43 with the per-object constant 1/(phong_alpha+1) precomputed (inv_exp).
__device__ __forceinline__ f3 hemisphere(const FrameParams &fp, Rng &rng, f3 n, float inv_exp) {
  const float cos_theta = pow_spec(rng_next(fp, rng), inv_exp);
  const float sin_theta = sqrtf(fmaf(-cos_theta, cos_theta, 1.0f));
  const float phi = kTwoPi * rng_next(fp, rng);
  float sphi, cphi;
  sincos_spec(phi, sphi, cphi);
  const float tx = cphi * sin_theta, ty = sphi * sin_theta, tz = cos_theta;
  const f3 helper = (fabsf(n.x) > 0.99f) ? mk3(0.0f, 0.0f, 1.0f) : mk3(1.0f, 0.0f, 0.0f);
  const f3 tangent = normalize3(cross3(n, helper));
  const f3 binormal = normalize3(cross3(n, tangent));
  return fma3(n, tz, fma3(binormal, ty, tangent * tx));
}

// frag.hlsl:8-12
__device__ __forceinline__ f3 sample_hdri(const FrameParams &fp, f3 d) {
  const float u = fmaf(atan2_spec(d.x, d.z), kInvTwoPi, 0.5f);
  const float v = fmaf(asin_spec(-d.y), kInvPi, 0.5f);
  const uint32_t i = texel_index(u, fp.hdri_w), j = texel_index(v, fp.hdri_h);
  const float4 t = fp.hdri[static_cast<size_t>(j) * fp.hdri_w + i];
  return mk3(t.x, t.y, t.z);
}

struct Hit {
  float t;
  int obj;    // original object index, -1 = miss
  int slot;   // sphere slot or plane slot
  bool plane;
};

// ray.hlsl:97-107 over the compiled scene.  GEOM_LDS: sphere records come from LDS.
template <bool GEOM_LDS>
__device__ __forceinline__ Hit intersect(const DeviceScene &sc, const float4 *lds_sph, f3 o, f3 d) {
  float best = 1000000.0f;
  int slot = -1;
  const uint32_t n = sc.n_sph;
  const float4 *sph = GEOM_LDS ? lds_sph : sc.sph;
  // One sphere of ray.hlsl:53-68.  q = discriminant/4 and dn = b/2 (exact power-of-two
  // scalings of the reference's b, c, discriminant), so t = -dn - sqrt(q) has the bits of
  // (-b - sqrt(disc)) / (2a).  A sphere can only be accepted when q >= 0 and dn < 0 (t > 0).
#define RT_SPHERE_PRE(g, dn, q, cand)                                  \
  const float nx##dn = o.x - g.x, ny##dn = o.y - g.y, nz##dn = o.z - g.z; \
  const float dn = fmaf(d.z, nz##dn, fmaf(d.y, ny##dn, d.x * nx##dn));    \
  const float q = fmaf(dn, dn, -(fmaf(nz##dn, nz##dn, fmaf(ny##dn, ny##dn, nx##dn * nx##dn)) - g.w)); \
  const bool cand = (q >= 0.0f) & (dn < 0.0f);
#define RT_SPHERE_POST(idx, dn, q, cand)                               \
  if (__builtin_amdgcn_ballot_w64(cand) != 0ull) {                     \
    const float t = -dn - sqrtf(q);                                    \
    const bool acc = cand & (t >= kEpsilon) & (t < best);              \
    best = acc ? t : best;                                             \
    slot = acc ? static_cast<int>(idx) : slot;                         \
  }
  uint32_t i = 0;
  for (; i + 4 <= n; i += 4) {
    const float4 g0 = sph[i], g1 = sph[i + 1], g2 = sph[i + 2], g3 = sph[i + 3];
    RT_SPHERE_PRE(g0, dn0, q0, c0)
    RT_SPHERE_PRE(g1, dn1, q1, c1)
    RT_SPHERE_PRE(g2, dn2, q2, c2)
    RT_SPHERE_PRE(g3, dn3, q3, c3)
    if (__builtin_amdgcn_ballot_w64(c0 | c1 | c2 | c3) != 0ull) {
      RT_SPHERE_POST(i, dn0, q0, c0)
      RT_SPHERE_POST(i + 1, dn1, q1, c1)
      RT_SPHERE_POST(i + 2, dn2, q2, c2)
      RT_SPHERE_POST(i + 3, dn3, q3, c3)
    }
  }
  for (; i < n; ++i) {
    const float4 g0 = sph[i];
    RT_SPHERE_PRE(g0, dn0, q0, c0)
    RT_SPHERE_POST(i, dn0, q0, c0)
  }
#undef RT_SPHERE_PRE
#undef RT_SPHERE_POST
  Hit h;
  h.t = best;
  h.slot = slot;
  h.plane = false;
  h.obj = (slot >= 0) ? sc.sph_obj[slot] : -1;
  for (uint32_t j = 0; j < sc.n_pl; ++j) {
    const float4 a = sc.pl_a[j], b = sc.pl_b[j];
    const f3 c = mk3(a.x, a.y, a.z), nrm = mk3(b.x, b.y, b.z);
    const int idx = static_cast<int>(f2u(b.w));
    const float den = dot3(d, nrm);
    if (fabsf(den) < kEpsilon) continue;
    const float num = dot3(c - o, nrm);
    const float t = num / den;
    const f3 p = fma3(d, t, o);
    const bool outside = (fabsf(p.x - c.x) > a.w) || (fabsf(p.y - c.y) > a.w) || (fabsf(p.z - c.z) > a.w);
    // lexicographic (distance, index): an equal distance only wins with the lower index
    const bool closer = (t < h.t) | ((t == h.t) & (idx < h.obj));
    if (!outside & closer & (t >= kEpsilon)) {
      h.t = t;
      h.obj = idx;
      h.slot = static_cast<int>(j);
      h.plane = true;
    }
  }
  return h;
}

template <bool GEOM_LDS>
__global__ void __launch_bounds__(kBlock) render_tiles(DeviceScene sc, FrameParams fp) {
  extern __shared__ float4 lds_sph[];
  if (GEOM_LDS) {
    for (uint32_t i = threadIdx.x; i < sc.n_sph; i += kBlock) lds_sph[i] = sc.sph[i];
    __syncthreads();
  }
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  const uint32_t tiles_x = (fp.width + kTile - 1) / kTile;
  const uint32_t tx = wave % tiles_x, ty = wave / tiles_x;
  const uint32_t x = tx * kTile + (lane & 7u);
  const uint32_t lrow = ty * kTile + (lane >> 3);
  const bool valid = (x < fp.width) & (lrow < fp.local_rows);
  uint32_t seg = 0, rough = 0;
  if (valid) {
    // local row -> global row of the round-robin strip partition
    const uint32_t y = ((lrow / fp.strip_rows) * fp.world + fp.rank) * fp.strip_rows + lrow % fp.strip_rows;
    const size_t p = static_cast<size_t>(lrow) * fp.width + x;
    // frag.hlsl:51-57 (D14: SV_POSITION.xy = pixel centre)
    const float pjx = (static_cast<float>(x) + 0.5f) + fp.jitter_x;
    const float pjy = (static_cast<float>(y) + 0.5f) + fp.jitter_y;
    const float cjx = pjx / static_cast<float>(fp.width), cjy = pjy / static_cast<float>(fp.height);
    Rng rng{42.0f, cjx, cjy};
    // ray.hlsl:31-37
    f3 pw = fma3(fp.cam_right, cjx * fp.vp_width, fp.top_left);
    pw = fma3(fp.cam_neg_up, cjy * fp.vp_height, pw);
    f3 o = fp.cam_pos;
    f3 d = normalize3(pw - fp.cam_pos);
    f3 energy = mk3(1.0f, 1.0f, 1.0f);
    f3 result = mk3(0.0f, 0.0f, 0.0f);
    // frag.hlsl:32-46
    for (uint32_t i = 0; i < fp.reflection_limit; ++i) {
      const Hit h = intersect<GEOM_LDS>(sc, lds_sph, o, d);
      if (fp.hits) fp.hits[p * fp.reflection_limit + i] = h.obj;
      seg += 1;
      if (h.obj < 0) {  // frag.hlsl:14-17: miss -> skybox, energy 0, loop ends (|0| < EPSILON)
        result = fma3(energy, sample_hdri(fp, d), result);
        break;
      }
      const f3 pos = fma3(d, h.t, o);
      f3 nrm;
      if (h.plane) {
        const float4 b = sc.pl_b[h.slot];
        nrm = mk3(b.x, b.y, b.z);  // ray.hlsl:43-45: stored normal, not renormalised
      } else {
        const float4 g = GEOM_LDS ? lds_sph[h.slot] : sc.sph[h.slot];
        nrm = normalize3(pos - mk3(g.x, g.y, g.z));
      }
      const float4 m0 = sc.mat0[h.obj], m1 = sc.mat1[h.obj];
      o = fma3(nrm, 0.001f, pos);                   // frag.hlsl:21
      const float k = 2.0f * dot3(nrm, d);          // reflect(), frag.hlsl:23
      f3 nd = fma3(nrm, -k, d);
      if (m0.w != 0.0f) {                           // random.hlsl:26-28
        nd = hemisphere(fp, rng, nd, m1.w);
        rough += 1;
      }
      d = nd;
      const float cosine = saturate(dot3(nrm, d));
      result = fma3(energy, mk3(m1.x, m1.y, m1.z), result);  // D1: energy before the update
      energy = energy * (mk3(m0.x, m0.y, m0.z) * cosine);    // frag.hlsl:27
      if (length3(energy) < kEpsilon) break;                  // frag.hlsl:40-42
    }
    // frag.hlsl:60-65 + the Rgba8UnormSrgb store (D13).  At progressive_count 0 the history
    // weight is exactly 0, so it is not read.
    f3 prev = mk3(0.0f, 0.0f, 0.0f);
    if (fp.progressive_count != 0u) {
      if (fp.accum_mode == 1) {
        prev = mk3(fp.hist_f32[3 * p], fp.hist_f32[3 * p + 1], fp.hist_f32[3 * p + 2]);
      } else {
        const uint32_t w = fp.rgba8[p];
        prev = mk3(fp.srgb_lut[w & 255u], fp.srgb_lut[(w >> 8) & 255u], fp.srgb_lut[(w >> 16) & 255u]);
      }
    }
    const f3 mixed = fma3(result, fp.opacity, prev * fp.keep);
    fp.rgba8[p] = srgb8_encode_spec(mixed.x) | (srgb8_encode_spec(mixed.y) << 8) |
                  (srgb8_encode_spec(mixed.z) << 16) | 0xff000000u;
    if (fp.rgbf32) {
      fp.rgbf32[3 * p] = mixed.x;
      fp.rgbf32[3 * p + 1] = mixed.y;
      fp.rgbf32[3 * p + 2] = mixed.z;
    }
    if (fp.hist_f32 && fp.accum_mode == 1) {
      fp.hist_f32[3 * p] = mixed.x;
      fp.hist_f32[3 * p + 1] = mixed.y;
      fp.hist_f32[3 * p + 2] = mixed.z;
    }
    if (fp.segments) fp.segments[p] = seg;
    if (fp.hits)  // iterations that never ran are marked -2
      for (uint32_t i = seg; i < fp.reflection_limit; ++i) fp.hits[p * fp.reflection_limit + i] = -2;
  }
  // one atomic per wave for the frame totals
  uint32_t packed_lo = seg, packed_hi = rough;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    packed_lo += __shfl_xor(packed_lo, off, 64);
    packed_hi += __shfl_xor(packed_hi, off, 64);
  }
  if (lane == 0 && packed_lo != 0u) {
    atomicAdd(&fp.counters[0], static_cast<unsigned long long>(packed_lo));
    if (packed_hi) atomicAdd(&fp.counters[1], static_cast<unsigned long long>(packed_hi));
  }
}

__global__ void math_unary_kernel(int op, const float *x, float *y, size_t n) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s, c, r;
  switch (op) {
    case 0: sincos_spec(x[i], s, c); r = s; break;
    case 1: sincos_spec(x[i], s, c); r = c; break;
    case 2: r = tan_spec(x[i]); break;
    case 3: r = asin_spec(x[i]); break;
    case 4: r = log2_spec(x[i]); break;
    case 5: r = exp2_spec(x[i]); break;
    default: r = sqrtf(x[i]); break;
  }
  y[i] = r;
}

__global__ void math_binary_kernel(int op, const float *a, const float *b, float *y, size_t n) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  y[i] = op == 0 ? atan2_spec(a[i], b[i]) : (op == 1 ? pow_spec(a[i], b[i]) : a[i] / b[i]);
}

}  // namespace

int launch_render(const DeviceScene &sc, const FrameParams &fp, const LaunchConfig &lc, hipStream_t stream) {
  const uint32_t tiles_x = (fp.width + kTile - 1) / kTile;
  const uint32_t tiles_y = (fp.local_rows + kTile - 1) / kTile;
  const uint64_t waves = static_cast<uint64_t>(tiles_x) * tiles_y;
  if (waves == 0) return 0;
  const uint32_t blocks = static_cast<uint32_t>((waves + (kBlock / 64) - 1) / (kBlock / 64));
  if (lc.variant == 1) {
    hipLaunchKernelGGL(render_tiles<false>, dim3(blocks), dim3(kBlock), 0, stream, sc, fp);
  } else {
    const size_t lds = static_cast<size_t>(sc.n_sph) * sizeof(float4);
    hipLaunchKernelGGL(render_tiles<true>, dim3(blocks), dim3(kBlock), lds, stream, sc, fp);
  }
  return static_cast<int>(hipGetLastError());
}

int launch_math_unary(int op, const float *x, float *y, size_t n, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(math_unary_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, stream, op, x, y, n);
  return static_cast<int>(hipGetLastError());
}

int launch_math_binary(int op, const float *a, const float *b, float *y, size_t n, hipStream_t stream) {
  if (n == 0) return 0;
  hipLaunchKernelGGL(math_binary_kernel, dim3(static_cast<uint32_t>((n + 255) / 256)), dim3(256), 0, stream, op, a, b, y, n);
  return static_cast<int>(hipGetLastError());
}

}  // namespace rt355
