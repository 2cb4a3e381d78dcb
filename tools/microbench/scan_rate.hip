// Microbenchmark: the sphere-scan loop of rt_kernels.hip in isolation, in four forms, on realistic secondary rays.
//   0  LDS broadcast reads (ds_read_b128), 13 VALU per sphere                       (= scan_spheres today)
//   1  LDS, two spheres per instruction with packed FP32 (v_pk_add/mul/fma_f32)
//   2  sphere records as SGPR operands (s_load from the scalar cache), 13 VALU per sphere
//   3  SGPR operands + packed FP32
// All four must produce the same (key, slot) per ray — checked on the host — so this also tells whether the
// packed forms are bit-identical on the hardware.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o scan_rate scan_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr uint32_t kEpsBits = 0x358637bdu;
constexpr uint32_t kNoHitKey = 0x49742400u - kEpsBits;

__device__ __forceinline__ float sqrt_rn(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sd = __uint_as_float(__float_as_uint(s) - 1u);
  const float su = __uint_as_float(__float_as_uint(s) + 1u);
  const float rd = fmaf(-sd, s, x);
  const float ru = fmaf(-su, s, x);
  float r = (rd <= 0.0f) ? sd : s;
  r = (ru > 0.0f) ? su : r;
  const bool tiny = (__float_as_uint(x) - 1u) < (0x0f800000u - 1u);
  if (__builtin_amdgcn_ballot_w64(tiny) != 0ull) r = tiny ? sqrtf(x) : r;
  return r;
}

struct Ray {
  float ox, oy, oz, dx, dy, dz;
};

#define POST(idx, dn, q, cand)                                         \
  if (cand != 0ull) {                                                  \
    const uint32_t key = __float_as_uint(-dn - sqrt_rn(q)) - kEpsBits; \
    slot = (key < best) ? static_cast<int>(idx) : slot;                \
    best = min(best, key);                                             \
  }

// ---- 0 / 2: one sphere per instruction; `sph` points to LDS (0) or to global memory read with uniform addresses (2)
template <bool SCALAR>
__device__ __forceinline__ void scan_plain(const float4 *__restrict__ sph, uint32_t n, const Ray &r, uint32_t &best, int &slot) {
#define PRE(g, dn, q, cand)                                                                         \
  const float nx##dn = r.ox - g.x, ny##dn = r.oy - g.y, nz##dn = r.oz - g.z;                        \
  const float dn = fmaf(r.dz, nz##dn, fmaf(r.dy, ny##dn, r.dx * nx##dn));                           \
  const float q = fmaf(dn, dn, -(fmaf(nz##dn, nz##dn, fmaf(ny##dn, ny##dn, nx##dn * nx##dn)) - g.w)); \
  const unsigned long long cand = __builtin_amdgcn_fcmpf(fminf(q, -dn), 0.0f, 3);
  for (uint32_t i = 0; i < n; i += 4) {
    const float4 g0 = sph[i], g1 = sph[i + 1], g2 = sph[i + 2], g3 = sph[i + 3];
    PRE(g0, dn0, q0, c0)
    PRE(g1, dn1, q1, c1)
    PRE(g2, dn2, q2, c2)
    PRE(g3, dn3, q3, c3)
    if ((c0 | c1 | c2 | c3) != 0ull) {
      POST(i, dn0, q0, c0)
      POST(i + 1, dn1, q1, c1)
      POST(i + 2, dn2, q2, c2)
      POST(i + 3, dn3, q3, c3)
    }
  }
#undef PRE
}

// ---- 1 / 3: two spheres per instruction.  pairs[2k] = {cxA, cxB, cyA, cyB}, pairs[2k+1] = {czA, czB, r2A, r2B}
__device__ __forceinline__ void scan_packed(const float4 *__restrict__ pairs, uint32_t n, const Ray &r, uint32_t &best, int &slot) {
  const f2 ox = {r.ox, r.ox}, oy = {r.oy, r.oy}, oz = {r.oz, r.oz};
  const f2 dx = {r.dx, r.dx}, dy = {r.dy, r.dy}, dz = {r.dz, r.dz};
#define PREP(a, b, dn, q)                                                                               \
  const f2 nx##dn = ox - f2{a.x, a.y}, ny##dn = oy - f2{a.z, a.w}, nz##dn = oz - f2{b.x, b.y};          \
  const f2 dn = __builtin_elementwise_fma(dz, nz##dn, __builtin_elementwise_fma(dy, ny##dn, dx * nx##dn)); \
  const f2 nn##dn = __builtin_elementwise_fma(nz##dn, nz##dn, __builtin_elementwise_fma(ny##dn, ny##dn, nx##dn * nx##dn)); \
  const f2 q = __builtin_elementwise_fma(dn, dn, -(nn##dn - f2{b.z, b.w}));
  for (uint32_t i = 0; i < n; i += 4) {  // 4 spheres = 2 pairs per trip
    const float4 a0 = pairs[i], b0 = pairs[i + 1], a1 = pairs[i + 2], b1 = pairs[i + 3];
    PREP(a0, b0, dnA, qA)
    PREP(a1, b1, dnB, qB)
    const unsigned long long c0 = __builtin_amdgcn_fcmpf(fminf(qA.x, -dnA.x), 0.0f, 3);
    const unsigned long long c1 = __builtin_amdgcn_fcmpf(fminf(qA.y, -dnA.y), 0.0f, 3);
    const unsigned long long c2 = __builtin_amdgcn_fcmpf(fminf(qB.x, -dnB.x), 0.0f, 3);
    const unsigned long long c3 = __builtin_amdgcn_fcmpf(fminf(qB.y, -dnB.y), 0.0f, 3);
    if ((c0 | c1 | c2 | c3) != 0ull) {
      POST(i, dnA.x, qA.x, c0)
      POST(i + 1, dnA.y, qA.y, c1)
      POST(i + 2, dnB.x, qB.x, c2)
      POST(i + 3, dnB.y, qB.y, c3)
    }
  }
#undef PREP
}

template <int MODE>
__global__ void __launch_bounds__(256) scan_kernel(const float4 *__restrict__ sph, const float4 *__restrict__ pairs, uint32_t n,
                                                   const Ray *__restrict__ rays, uint32_t rays_per_lane, uint2 *out) {
  extern __shared__ float4 lds[];
  if (MODE == 0)
    for (uint32_t i = threadIdx.x; i < n; i += 256) lds[i] = sph[i];
  if (MODE == 1)
    for (uint32_t i = threadIdx.x; i < n; i += 256) lds[i] = pairs[i];
  __syncthreads();
  const uint32_t lane_global = blockIdx.x * 256 + threadIdx.x;
  uint32_t acc_key = 0, acc_slot = 0;
  Ray r = rays[static_cast<size_t>(lane_global) * rays_per_lane];
  for (uint32_t k = 0; k < rays_per_lane; ++k) {
    const Ray next = rays[static_cast<size_t>(lane_global) * rays_per_lane + (k + 1 < rays_per_lane ? k + 1 : k)];
    uint32_t best = kNoHitKey;
    int slot = -1;
    if (MODE == 0) scan_plain<false>(lds, n, r, best, slot);
    if (MODE == 1) scan_packed(lds, n, r, best, slot);
    if (MODE == 2) scan_plain<true>(sph, n, r, best, slot);
    if (MODE == 3) scan_packed(pairs, n, r, best, slot);
    acc_key = acc_key * 31u + best;
    acc_slot = acc_slot * 31u + static_cast<uint32_t>(slot);
    r = next;
  }
  out[lane_global] = make_uint2(acc_key, acc_slot);
}

static uint64_t rng_state = 0x1234567ull;
static float frand() {
  rng_state = rng_state * 6364136223846793005ull + 1442695040888963407ull;
  return static_cast<float>((rng_state >> 40) & 0xffffff) / 16777216.0f;
}

int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? atoi(argv[1]) : 64;             // spheres (multiple of 4)
  const uint32_t rays_per_lane = argc > 2 ? atoi(argv[2]) : 64;
  const int blocks_per_cu = argc > 3 ? atoi(argv[3]) : 6;
  std::vector<float4> sph(n), pairs(n);
  for (uint32_t i = 0; i < n; ++i) {
    const float rad = 3.0f + 5.0f * frand(), a = 6.2831853f * frand(), rr = 50.0f * sqrtf(frand());
    sph[i] = make_float4(rr * cosf(a), rad, rr * sinf(a), rad * rad);
  }
  for (uint32_t k = 0; k < n / 2; ++k) {
    const float4 A = sph[2 * k], B = sph[2 * k + 1];
    pairs[2 * k] = make_float4(A.x, B.x, A.y, B.y);
    pairs[2 * k + 1] = make_float4(A.z, B.z, A.w, B.w);
  }
  const uint32_t blocks = 256 * blocks_per_cu, lanes = blocks * 256;
  std::vector<Ray> rays(static_cast<size_t>(lanes) * rays_per_lane);
  for (auto &r : rays) {  // a reflection off a random sphere: origin just above its surface, direction in the normal's hemisphere
    const float4 s = sph[static_cast<uint32_t>(frand() * n) % n];
    float nx, ny, nz, l;
    do { nx = 2 * frand() - 1; ny = frand(); nz = 2 * frand() - 1; l = nx * nx + ny * ny + nz * nz; } while (l > 1.0f || l < 1e-3f);
    l = 1.0f / sqrtf(l); nx *= l; ny *= l; nz *= l;
    const float rad = sqrtf(s.w) * 1.001f;
    float dx, dy, dz;
    do { dx = 2 * frand() - 1; dy = 2 * frand() - 1; dz = 2 * frand() - 1; l = dx * dx + dy * dy + dz * dz; } while (l > 1.0f || l < 1e-3f || dx * nx + dy * ny + dz * nz < 0);
    l = 1.0f / sqrtf(l);
    r = Ray{s.x + nx * rad, s.y + ny * rad, s.z + nz * rad, dx * l, dy * l, dz * l};
  }
  float4 *d_sph, *d_pairs;
  Ray *d_rays;
  uint2 *d_out;
  hipMalloc(&d_sph, n * 16); hipMalloc(&d_pairs, n * 16);
  hipMalloc(&d_rays, rays.size() * sizeof(Ray)); hipMalloc(&d_out, lanes * 8);
  hipMemcpy(d_sph, sph.data(), n * 16, hipMemcpyHostToDevice);
  hipMemcpy(d_pairs, pairs.data(), n * 16, hipMemcpyHostToDevice);
  hipMemcpy(d_rays, rays.data(), rays.size() * sizeof(Ray), hipMemcpyHostToDevice);
  std::vector<uint2> ref(lanes), got(lanes);
  const char *names[] = {"LDS, 1 sphere/instr", "LDS, packed pairs", "SGPR, 1 sphere/instr", "SGPR, packed pairs"};
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep)
    for (int mode = 0; mode < 4; ++mode) {
      hipEventRecord(e0);
      for (int it = 0; it < 10; ++it) {
        if (mode == 0) hipLaunchKernelGGL(scan_kernel<0>, dim3(blocks), dim3(256), n * 16, 0, d_sph, d_pairs, n, d_rays, rays_per_lane, d_out);
        if (mode == 1) hipLaunchKernelGGL(scan_kernel<1>, dim3(blocks), dim3(256), n * 16, 0, d_sph, d_pairs, n, d_rays, rays_per_lane, d_out);
        if (mode == 2) hipLaunchKernelGGL(scan_kernel<2>, dim3(blocks), dim3(256), 0, 0, d_sph, d_pairs, n, d_rays, rays_per_lane, d_out);
        if (mode == 3) hipLaunchKernelGGL(scan_kernel<3>, dim3(blocks), dim3(256), 0, 0, d_sph, d_pairs, n, d_rays, rays_per_lane, d_out);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(got.data(), d_out, lanes * 8, hipMemcpyDeviceToHost);
      if (mode == 0) ref = got;
      size_t bad = 0;
      for (uint32_t i = 0; i < lanes; ++i) bad += got[i].x != ref[i].x || got[i].y != ref[i].y;
      const double tests = 10.0 * lanes * rays_per_lane * n;
      if (rep == 2)
        printf("%-22s n=%u  %7.3f ms  %6.1f G sphere tests/s  %6.2f Grays/s  SIMD-cycles per wave-sphere @2.4GHz %.2f  mismatches vs mode 0: %zu\n",
               names[mode], n, ms / 10, tests / (ms * 1e-3) / 1e9, tests / n / (ms * 1e-3) / 1e9,
               (ms * 1e-3 * 2.4e9 * 1024) / (tests / 64), bad);
    }
  return 0;
}
