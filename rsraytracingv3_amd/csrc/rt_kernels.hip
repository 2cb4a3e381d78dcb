// rt_kernels.hip — gfx950 kernels of the rt355 render path.
//
// One lane carries one pixel's path through frag.hlsl:48-66 (fs_main).  The sphere records of
// the compiled scene are staged into LDS once per workgroup and every ray segment scans them
// from there (wave-uniform address -> LDS broadcast, conflict free).  Closest-hit selection
// follows ray.hlsl:97-107 exactly (App. B D10: strict '<' in ascending object order ==
// lexicographic min of (distance, index)).
//
// Four launch shapes share the per-segment code (rt_set_option(RT_OPT_VARIANT); -1 = pick by size):
//   render_tiles       (0 / 1) one 8x8 pixel tile per wave, lanes idle once their path has ended;
//                      sphere records from LDS (0) or straight from global memory (1)
//   render_persistent  (2) resident waves pull 8x8 tiles from sharded device counters and REFILL
//                      lanes whose path ended with the next pixels (ballot + mbcnt rank)
//   render_blocks      (3) one tile per wave, but after every bounce the workgroup packs its
//                      surviving paths into as few waves as possible through LDS and ends the rest
//   primary_stream +   (4) one tile per wave for the primary segment only; surviving paths go through a queue in
//   bounce_stream      global memory to resident waves that refill finished lanes from it (sample-only launches)
//
// The square root of the sphere test is only evaluated when some lane of the wave can still
// accept the sphere (ballot over "discriminant >= 0 and the sphere is in front"): lanes the
// reference rejects never use the root, so skipping it is exact.
#include <hip/hip_runtime.h>

#ifndef RT_SHADE_SQRT_RN
#define RT_SHADE_SQRT_RN 1  // 1 = the shading code's square roots through sqrt_rn (below: 11 instructions, correctly rounded for
                            // every float, rt_selftest_sqrt) instead of hipcc's 16-instruction sqrtf expansion.  A/B on one
                            // device: +0.6 % (it cost 5 % while the kernel had no register to spare)
#endif
#if RT_SHADE_SQRT_RN && defined(__HIP_DEVICE_COMPILE__)
namespace rt355 {
namespace {
__device__ __forceinline__ float sqrt_rn(float x);
__device__ __forceinline__ float rcp_of_root(float s);
}
}  // namespace rt355
#define RT_SQRTF(x) sqrt_rn(x)
#define RT_RCP_OF_ROOT(s) rcp_of_root(s)
#endif
#include "rt_device.h"

static_assert(rt355::kSphBatch == 8, "the scan loop consumes 8 sphere records per trip");

namespace rt355 {

namespace {

#ifndef RT_PERSISTENT_WAVES
#define RT_PERSISTENT_WAVES 6  // waves per SIMD the persistent kernel's register budget is held to
#endif
#ifndef RT_BLOCKS_WAVES_BATCH
#define RT_BLOCKS_WAVES_BATCH 7   // waves per SIMD the re-packing shape is budgeted for in multi-frame launches: 72 VGPRs, no scratch
                                  // (round 2, A/B on one device: 6 / 7 / 8 waves 37.7 / 38.1 / 35.8 Grays/s; round 3, with the balanced
                                  // direction-mask scan: 5 / 6 / 7 / 8 waves 45.4 / 49.3 / 50.2 / 47.2 — at 7 it only fits since the
                                  // scan recomputes its lane-derived LDS addresses instead of keeping them across the bounce loop)
#endif
#ifndef RT_BLOCKS_WAVES_BATCH_WIDE
#define RT_BLOCKS_WAVES_BATCH_WIDE 7  // ... for scenes whose direction-mask sets take two words (65..128 spheres, the reference's 80
                                      // slots): 6 / 7 waves 44.4 / 43.6 Grays/s before the tail spill and the phase priorities, 48.5 / 49.0 with them
#endif
#ifndef RT_BLOCKS_WAVES_LISTS
#define RT_BLOCKS_WAVES_LISTS 6   // ... for large scenes (candidate lists; nothing staged into LDS)
#endif
#ifndef RT_LIST_TRIPS2
#define RT_LIST_TRIPS2 1  // candidate-list scan: two (ray, sphere) pairs per lane and trip, their dependent reads in flight together
#endif
#ifndef RT_LIST_NT_LOADS
#define RT_LIST_NT_LOADS 0  // 1 = candidate-list entries loaded with the non-temporal hint: A/B on one device 32.4 against 36.1 Grays/s at 1 024 spheres —
                            // neighbouring rays share entries, and a streamed line is gone when the next one asks
#endif
#ifndef RT_LIST_LDS_COUNTER
#define RT_LIST_LDS_COUNTER 0  // pair positions from one LDS atomic per lane instead of a DPP prefix sum
#endif
#ifndef RT_LIST_EARLY_X
#define RT_LIST_EARLY_X 1  // kernels of large scenes: the shading's random() texels requested around the scan
#endif
#ifndef RT_LAZY_OBJ
#define RT_LAZY_OBJ 1     // sample-only kernels of large scenes: a sphere hit's object index is fetched only to break a distance tie with a plane
#endif
#ifndef RT_BLOCKS_WAVES_SINGLE_LISTS
#define RT_BLOCKS_WAVES_SINGLE_LISTS 6  // ... their sample-only single-frame launches
#endif
#ifndef RT_BLOCKS_WAVES_SINGLE
#define RT_BLOCKS_WAVES_SINGLE 5  // ... in sample-only single-frame launches, several of which are in flight (register count held at the
                                  // 5-wave step).  5 / 6 / 7 waves, one frame per launch at 8 bounces: 47.2 / 44.8 / 46.3 Grays/s before the
                                  // phase priorities, 47.7 / 44.8 / 48.5 with them — but 7 against 5 by bounce limit 1 / 2 / 3 / 5 / 8:
                                  // -14 % / +1.8 % / -1.3 % / -0.3 % / +1.6 % (profiles/r03_ab_waves_with_priorities.txt): level but for the loss
#endif
#ifndef RT_BLOCKS_WAVES_FULL
#define RT_BLOCKS_WAVES_FULL 5    // ... and in launches with per-frame outputs (history in the kernel, diagnostics: 91 registers)
#endif
#ifndef RT_X2_PREFETCH
#define RT_X2_PREFETCH 0  // 1 = request the second random() texel during the scan, 0 = when the hit turns out rough (A/B on one
                          // device: 0 is +0.3 % since the kernel is issue-bound — the address arithmetic of a texel most
                          // iterations do not need costs more than its latency)
#endif
#ifndef RT_DEFER_SKY_BLOCKS
#define RT_DEFER_SKY_BLOCKS 1  // render_blocks / render_persistent: 1 = skybox lookup (atan2 + asin + texel, ~75 instructions the
                               // whole wave executes) in path_end, with the other finished lanes, 0 = at the miss after every
                               // bounce.  A/B on one device: +1.6 % (it cost 4 % while the kernel had no register to spare)
#endif
#ifdef RT_AB_SHAPES
constexpr int kBlock = 256;  // threads per workgroup of the resident-wave shape
#endif
#ifndef RT_TILES_BLOCK
#define RT_TILES_BLOCK 256
#endif
constexpr int kTileBlock = RT_TILES_BLOCK;  // threads per workgroup of render_tiles (independent waves sharing the staged scene)
constexpr int kTile = 8;  // 8x8 pixels per tile / wave
#ifndef RT_BLOCK_NW
#define RT_BLOCK_NW 2  // waves (8x8 tiles) per workgroup of the re-packing shape.  A/B on one device, headline / one
                       // frame per launch in Grays/s: 4 waves 31.8 / 32.0, 2 waves 33.4 / 30.8 at 6 waves per SIMD and
                       // 32.5 / 33.1 at 5, 1 wave 32.7 / 30.5, 8 waves 27.0 / 27.5 — hence 2 waves, and the register
                       // budget by launch kind
#endif
#ifndef RT_PLANES_AFTER_SCAN
#define RT_PLANES_AFTER_SCAN 0
#endif
#ifndef RT_MASK_PIPELINE
#define RT_MASK_PIPELINE 1  // direction-mask scans: request the next candidate's record before the current one is tested
#endif
#ifndef RT_BLOCK_NW_LISTS
#define RT_BLOCK_NW_LISTS 1  // ... of the kernels of large scenes (candidate lists, nothing staged): ONE wave.  Two waves share a staged scene
                             // image and re-pack their paths; without the image the count barrier and the re-packing only cost (a fifth
                             // of the waves' time: a wave that needs a second round of lookups holds its partner at the barrier), and the
                             // tail spill takes the thinned-out wave's paths anyway.  A/B on one device, 1 024 spheres: 37.4 against 35.8
                             // Grays/s (at 64 spheres, staged: 48.9 against 54.0 — hence two there)
#endif
constexpr uint32_t xchg_paths(uint32_t nw) { return (nw - 1u) * 64u; }  // paths the exchange buffer holds: re-packing needs them to fit one wave fewer
#ifndef RT_BALANCED_SCAN
#define RT_BALANCED_SCAN 1  // direction-mask scans of the re-packing shape: 1 = the wave's (ray, candidate) pairs are dealt evenly over
                            // its lanes through LDS instead of every lane walking its own set (trips = the wave's heaviest lane)
#endif
// Per-wave LDS scratch of the balanced direction-mask scan: 64 rays (float4 + float2), 64 best hits (u64), kPairCap pairs (u16).
constexpr uint32_t kPairCap = 512;
constexpr uint32_t kScanScratchBytes = 64u * 16u + 64u * 8u + 64u * 8u + kPairCap * 2u;  // 3072
// ... of the candidate-LIST scan of large scenes (rt_dirmask.h "Candidate lists"): 64 rays, 64 best hits, the 64 lanes' 32-byte
// entries as loaded, and up to 64 x 15 (lane, position) pairs
constexpr uint32_t kListPairCap = 64u * kDirListSlots;
constexpr uint32_t kListScratchBytes = 64u * 16u + 64u * 8u + 64u * 8u + 64u * 32u + kListPairCap * 2u;  // 6016
constexpr uint32_t xchg_bytes(uint32_t nw) { return xchg_paths(nw) * 5u * 16u; }
// With two waves per workgroup the scratch may share the exchange buffer's LDS: the buffer is only written after a count
// barrier that both waves reach after their scans, and once it has been used a single wave is left.  Other workgroup sizes
// keep the two apart.
constexpr bool scratch_in_xchg(uint32_t nw) { return nw <= 2u; }
constexpr uint32_t blocks_lds_extra(uint32_t nw, uint32_t scratch_bytes) {
  return scratch_in_xchg(nw) ? (xchg_bytes(nw) > nw * scratch_bytes ? xchg_bytes(nw) : nw * scratch_bytes) : xchg_bytes(nw) + nw * scratch_bytes;
}
constexpr uint32_t kBlocksLdsExtra = !RT_BALANCED_SCAN ? xchg_bytes(RT_BLOCK_NW) : blocks_lds_extra(RT_BLOCK_NW, kScanScratchBytes);
constexpr uint32_t kBlocksLdsExtraLists = blocks_lds_extra(RT_BLOCK_NW_LISTS, kListScratchBytes);

// Correctly rounded sqrt for x >= 2^-96 (also +0, +inf, NaN): v_sqrt_f32 is within 1 ulp, so
// the IEEE result is one of {s-1ulp, s, s+1ulp}; the signs of the two fused residuals pick it
// (the same selection hipcc's own expansion makes, without its pre-scaling of small inputs).
// Below 2^-96 the residuals can underflow, so those inputs take hipcc's scaled expansion behind
// a wave-uniform branch that is essentially never taken (0 < bits(x) < bits(2^-96) as one
// unsigned compare).  Checked against the IEEE result for EVERY float by rt_selftest_sqrt.
__device__ __forceinline__ float sqrt_rn(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sd = __uint_as_float(__float_as_uint(s) - 1u);
  const float su = __uint_as_float(__float_as_uint(s) + 1u);
  const float rd = fmaf(-sd, s, x);
  const float ru = fmaf(-su, s, x);
  float r = (rd <= 0.0f) ? sd : s;
  r = (ru > 0.0f) ? su : r;
  const bool tiny = (__float_as_uint(x) - 1u) < (0x0f800000u - 1u);
  if (__builtin_amdgcn_ballot_w64(tiny) != 0ull) {
    asm volatile("; rare path: sqrt of a value below 2^-96" ::: "memory");  // keeps this a real branch
    r = tiny ? sqrtf(x) : r;
  }
  return r;
}

// 1.0f / s, correctly rounded, for the arguments normalize3 has: s is a correctly rounded square root, i.e. +0, +inf, a
// NaN, or a normal number in [2^-75, 2^64].  v_rcp_f32 followed by ONE Newton step in fused arithmetic equals the IEEE
// quotient for every normal s with 2^-126 <= |s| <= 2^126 (tools/microbench/rcp_exhaustive.hip sweeps every float: 0
// mismatches; rt_selftest_sqrt repeats the sweep for this function); zero, infinity, NaN and denormals take hipcc's
// division behind a wave-uniform branch that real scenes never enter.  4 instructions instead of 11.
__device__ __forceinline__ float rcp_of_root(float s) {
  const float r0 = __builtin_amdgcn_rcpf(s);
  float r = fmaf(fmaf(-s, r0, 1.0f), r0, r0);
  const bool special = __builtin_amdgcn_classf(s, 0x2f7);  // anything but a +-normal number
  if (__builtin_amdgcn_ballot_w64(special) != 0ull) {
    asm volatile("; rare path: reciprocal of 0 / inf / NaN / a denormal" ::: "memory");  // keeps this a real branch
    r = special ? 1.0f / s : r;
  }
  return r;
}

// Diagnostic build only (make prof: -DRT_PROFILE_PHASES): per-wave s_memtime deltas of the phases of
// render_blocks, summed into words 4.. of the launch's counter shard.  Compiles to nothing otherwise.
#ifdef RT_PROFILE_PHASES
#define RT_PH_DECL unsigned long long ph_t = __builtin_amdgcn_s_memtime(), ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define RT_PH_MARK_RAW(k)                                            \
  {                                                                  \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();    \
    ph_acc[k] += now_ - ph_t;                                        \
    ph_t = now_;                                                     \
  }
#ifdef RT_PH_LISTS  // (diagnostic build of the candidate-list scan's own stamps: slots 0..6 are its steps, 7 everything else)
#define RT_PH_MARK(k) RT_PH_MARK_RAW(7)
#define RT_PHL(k) RT_PH_MARK_RAW(k)
#define RT_PHL_WAIT __builtin_amdgcn_s_waitcnt(0);
#else
#define RT_PH_MARK(k) RT_PH_MARK_RAW(k)
#define RT_PHL(k)
#define RT_PHL_WAIT
#endif
#define RT_PH_FLUSH(shard)                                                                         \
  if ((threadIdx.x & 63u) == 0)                                                                    \
    for (int k_ = 0; k_ < 8; ++k_)                                                                 \
      if (ph_acc[k_]) atomicAdd(&(shard)[4 + k_], ph_acc[k_]);
#define RT_PH_PARAMS , unsigned long long &ph_t, unsigned long long (&ph_acc)[8]
#define RT_PH_ARGS , ph_t, ph_acc
#else
#define RT_PH_DECL
#define RT_PH_MARK(k)
#define RT_PHL(k)
#define RT_PHL_WAIT
#define RT_PH_FLUSH(shard)
#define RT_PH_PARAMS
#define RT_PH_ARGS
#endif

// Diagnostic build only (make events: -DRT_PROFILE_EVENTS; its atomics slow the kernel several times, so it is a build of
// its own): event counts since the last rt_get_counter_words — [0] spheres tested by full scans (per wave), [1] of which
// the wave took the root + closest-hit update, [2] sphere steps of cooperative scans, [3] spheres tested by listed scans,
// [4] direction-mask scans (per wave), [5] their loop trips (a trip = one sphere per lane still in the loop), [6] candidates
// over all lanes, [7] rays (live lanes) of those scans.
#ifdef RT_PROFILE_EVENTS
__device__ unsigned long long g_prof_cnt[8];
// one count per wave from inside divergent code: the first lane that is active here adds it
#define RT_CNT_ANY(k, n)                                                                                                   \
  {                                                                                                                        \
    if ((threadIdx.x & 63u) == static_cast<uint32_t>(__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))))                   \
      atomicAdd(&g_prof_cnt[k], static_cast<unsigned long long>(n));                                                       \
  }
#define RT_CNT_LANE(k, n) atomicAdd(&g_prof_cnt[k], static_cast<unsigned long long>(n));
#define RT_CNT(k, n)                                                                             \
  {                                                                                              \
    if ((threadIdx.x & 63u) == 0) atomicAdd(&g_prof_cnt[k], static_cast<unsigned long long>(n)); \
  }
#else
#define RT_CNT(k, n)
#define RT_CNT_ANY(k, n)
#define RT_CNT_LANE(k, n)
#endif

// The root of the sphere test, t = -dn - sqrt(q): sqrt_rn without its small-input branch.  For q >= 2^-96 (and +0)
// this IS the correctly rounded root.  For 0 < q < 2^-96 the residuals may underflow and the result is only some value
// r in [0, 2^-47] (one of v_sqrt_f32's result and its two neighbours) — which gives the same hit key as the exact root,
// itself below 2^-48: when -dn >= 2^-22 both are less than a quarter ulp of -dn and fl(-dn - r) = -dn either way; when
// -dn < 2^-22 then t <= -dn < EPSILON (1e-6) for every r >= 0 and the sphere is rejected either way (ray.hlsl:53-68
// accepts t >= EPSILON only).  Two instructions fewer per tested sphere than sqrt_rn; tools/fuzz_parity.py and the
// degenerate-sphere cases of tests/test_parity_gpu.py cover it.
__device__ __forceinline__ float sqrt_for_hit(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float sd = __uint_as_float(__float_as_uint(s) - 1u);
  const float su = __uint_as_float(__float_as_uint(s) + 1u);
  const float rd = fmaf(-sd, s, x);
  const float ru = fmaf(-su, s, x);
  float r = (rd <= 0.0f) ? sd : s;
  r = (ru > 0.0f) ? su : r;
  return r;
}
constexpr uint32_t kGridMaxDimDevice = 64;  // = kGridMaxDim of rt_grid.h

struct Rng {
  float seed, px, py;  // random.hlsl:1-2
};

// random.hlsl:4-11 in two halves so the texel can be requested long before it is needed.
// rng_texel: coord = (rng_pixel + rng_seed) % 1.0, nearest texel (both SampleLevel calls of the
// reference address this one texel).
__device__ __forceinline__ size_t rng_texel(const FrameParams &fp, const Rng &rng) {
  const float sx = rng.px + rng.seed, sy = rng.py + rng.seed;
  const float cx = sx - floorf(sx), cy = sy - floorf(sy);
  const uint32_t i = texel_index(cx, fp.rnd_wf, fp.rnd_w1f), j = texel_index(cy, fp.rnd_hf, fp.rnd_h1f);
  return static_cast<size_t>(j) * fp.rnd_w + i;
}
// rng_next_seed: rng_seed = (rng_seed + x * 532.3412) % 600.  The seed stays in [0, 1200) for texture
// values in [0, 1), where one conditional subtraction is the exact remainder; anything else takes fmodf.
__device__ __forceinline__ float rng_next_seed(float seed, float x) {
  float s = fmaf(x, 532.3412f, seed);
  if (s >= 0.0f && s < 1200.0f) {
    s = (s >= 600.0f) ? s - 600.0f : s;
  } else {
    s = fmodf(s, 600.0f);
  }
  return s;
}

// random.hlsl:13-43 with the per-object constant 1/(phong_alpha+1) precomputed (inv_exp).  x1 and x2 are the
// two random() values, both requested by the caller during the object scan: x1's texel depends only on the
// RNG state at the start of the segment, x2's on that state advanced by x1 (= seed1).
__device__ __forceinline__ f3 hemisphere(Rng &rng, f3 n, float inv_exp, float x1, float x2, float seed1) {
  rng.seed = rng_next_seed(seed1, x2);
  const float cos_theta = pow_spec(x1, inv_exp);
  // 1 - cos^2 is +0 or at least 2^-24 (cos_theta is a float <= 1): the branch-free root is the correctly rounded one
  const float sin_theta = sqrt_for_hit(fmaf(-cos_theta, cos_theta, 1.0f));
  const float phi = kTwoPi * x2;
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
  const uint32_t i = texel_index(u, fp.hdri_wf, fp.hdri_w1f), j = texel_index(v, fp.hdri_hf, fp.hdri_h1f);
  const float4 t = fp.hdri[static_cast<size_t>(j) * fp.hdri_w + i];
  return mk3(t.x, t.y, t.z);
}

// Where a kernel reads the scene image from (rt_device.h): LDS for what it staged, global memory for the rest.
struct SceneView {
  const float4 *sph, *smat0, *smat1, *pl_a, *pl_b, *pmat0, *pmat1;
  const int *sph_obj;
  uint32_t n_sph_pad, n_pl;
  const uint32_t *grid_cells;  // cell_start[n_cells + 1], or null
  const uint16_t *grid_items;
  const DirMaskParams *dm;     // candidate-sphere masks for secondary rays (rt_dirmask.h); enabled == 0: none
};
// `sph_base`: record 0 of the sphere array; `rest_base`: record 0 of an image that holds everything else
// (the two are the same pointer when the whole image was staged).
__device__ __forceinline__ SceneView make_view(const DeviceScene &sc, const float4 *sph_base, const float4 *rest_base) {
  SceneView v;
  v.sph = sph_base;
  v.smat0 = rest_base + sc.off_smat0();
  v.smat1 = rest_base + sc.off_smat1();
  v.pl_a = rest_base + sc.off_pl_a();
  v.pl_b = rest_base + sc.off_pl_b();
  v.pmat0 = rest_base + sc.off_pmat0();
  v.pmat1 = rest_base + sc.off_pmat1();
  v.sph_obj = reinterpret_cast<const int *>(rest_base + sc.off_sph_obj());
  v.n_sph_pad = sc.n_sph_pad;
  v.n_pl = sc.n_pl;
  v.grid_cells = nullptr;
  v.grid_items = nullptr;
  v.dm = &sc.dm;
  return v;
}

struct Hit {
  float t;
  int obj;    // original object index, -1 = miss
  int slot;   // sphere slot or plane slot
  bool plane;
};

// Closest sphere hit as an unsigned key: for EPSILON <= t < best (both positive floats) the IEEE bit
// patterns order like the values, and anything else (t < EPSILON, negative, NaN from a negative
// discriminant) wraps to a key >= the current one.  key = bits(t) - bits(EPSILON).
constexpr uint32_t kEpsBits = 0x358637bdu;              // 1e-6f
constexpr uint32_t kNoHitKey = 0x49742400u - kEpsBits;  // 1000000.0f, the initial hit.distance (ray.hlsl:98)

struct SphereHit {
  uint32_t key;  // bits(t) - bits(EPSILON), kNoHitKey = none
  int slot;      // sphere slot, -1 = none
};

// The sphere part of ray.hlsl:97-107, one ray per lane (call with the lanes that carry a ray).
// Records [begin, end) (multiples of 8; up to end + 3 is read), continuing from `acc`.
__device__ __forceinline__ SphereHit scan_spheres(const float4 *__restrict__ sph, uint32_t begin, uint32_t end, f3 o, f3 d,
                                                  SphereHit acc = SphereHit{kNoHitKey, -1}) {
  uint32_t best_key = acc.key;
  int slot = acc.slot;
  // One sphere of ray.hlsl:53-68.  q = discriminant/4 and dn = b/2 (exact power-of-two scalings of
  // the reference's b, c, discriminant), so t = -dn - sqrt(q) has the bits of (-b - sqrt(disc))/(2a).
  // A sphere can only be accepted when q >= 0 and dn < 0 (then t > 0); min(q, -dn) >= 0 is a
  // one-compare superset of that, good enough to decide whether the wave needs the root at all.
#define RT_SPHERE_PRE(g, dn, q, cand)                                                                  \
  const float nx##dn = o.x - g.x, ny##dn = o.y - g.y, nz##dn = o.z - g.z;                              \
  const float dn = fmaf(d.z, nz##dn, fmaf(d.y, ny##dn, d.x * nx##dn));                                 \
  const float q = fmaf(dn, dn, -(fmaf(nz##dn, nz##dn, fmaf(ny##dn, ny##dn, nx##dn * nx##dn)) - g.w)); \
  const unsigned long long cand = __builtin_amdgcn_fcmpf(fminf(q, -dn), 0.0f, 3 /* OGE */);
#define RT_SPHERE_POST(idx, dn, q, cand)                                     \
  if (cand != 0ull) {                                                        \
    RT_CNT(1, 1)                                                             \
    const uint32_t key = __float_as_uint(-dn - sqrt_for_hit(q)) - kEpsBits;       \
    slot = (key < best_key) ? static_cast<int>(idx) : slot;                  \
    best_key = min(best_key, key);                                           \
  }
#define RT_SPHERE_BATCH(a0, a1, a2, a3, base)                            \
  {                                                                      \
    RT_SPHERE_PRE(a0, dn0, q0, c0)                                       \
    RT_SPHERE_PRE(a1, dn1, q1, c1)                                       \
    RT_SPHERE_PRE(a2, dn2, q2, c2)                                       \
    RT_SPHERE_PRE(a3, dn3, q3, c3)                                       \
    if ((c0 | c1 | c2 | c3) != 0ull) {                                   \
      RT_SPHERE_POST((base), dn0, q0, c0)                                \
      RT_SPHERE_POST((base) + 1, dn1, q1, c1)                            \
      RT_SPHERE_POST((base) + 2, dn2, q2, c2)                            \
      RT_SPHERE_POST((base) + 3, dn3, q3, c3)                            \
    }                                                                    \
  }
  // Two batches of 4 per trip, each computed while the next one's LDS reads are in flight
  // (n_sph_pad is a multiple of 8 plus 8 readable never-hit records).
  RT_CNT(0, end - begin)
  float4 g0 = sph[begin], g1 = sph[begin + 1], g2 = sph[begin + 2], g3 = sph[begin + 3];
  for (uint32_t i = begin; i < end; i += 8) {
    const float4 h0 = sph[i + 4], h1 = sph[i + 5], h2 = sph[i + 6], h3 = sph[i + 7];
    RT_SPHERE_BATCH(g0, g1, g2, g3, i)
    g0 = sph[i + 8];
    g1 = sph[i + 9];
    g2 = sph[i + 10];
    g3 = sph[i + 11];
    RT_SPHERE_BATCH(h0, h1, h2, h3, i + 4)
  }
#undef RT_SPHERE_BATCH
#undef RT_SPHERE_PRE
#undef RT_SPHERE_POST
  return SphereHit{best_key, slot};
}

// Wave-cooperative sphere scan for waves with few live rays (<= 32 of 64 lanes): G = 2, 4, 8 or 16
// lanes work for one ray, each testing every G-th sphere, and the closest hit is selected across the
// G lanes with wavefront cross-lane operations — a lexicographic (key, slot) min over xor-butterfly
// partners, which is exactly the reference's "strict '<' in ascending index" rule (App. B D10).
// Rays are handed to lane groups and results handed back with ds_permute/ds_bpermute; every lane of
// the wave must call this (full EXEC), `alive` says which lanes own a ray.  A 16-ray wave costs 18
// sphere steps instead of 72.
__device__ __forceinline__ SphereHit scan_spheres_coop(const float4 *__restrict__ sph, uint32_t n_sph_pad, f3 o, f3 d,
                                                       bool alive, unsigned long long live, uint32_t n_live_rays) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t shift = n_live_rays <= 4u ? 4u : (n_live_rays <= 8u ? 3u : (n_live_rays <= 16u ? 2u : 1u));  // log2(G), wave-uniform
  const uint32_t G = 1u << shift;
  // rank of each live lane among the live lanes; dead lanes get the ranks after them, so that
  // "rank -> lane" is a permutation of 0..63 and ds_permute never collides
  const unsigned long long dead = ~live;
  const uint32_t live_below = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(live >> 32),
                                                        __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(live), 0u));
  const uint32_t dead_below = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(dead >> 32),
                                                        __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(dead), 0u));
  const uint32_t rank = alive ? live_below : n_live_rays + dead_below;
  const int owner = __builtin_amdgcn_ds_permute(static_cast<int>(rank << 2), static_cast<int>(lane));  // lane k: owner of ray k
  const uint32_t ray = lane >> shift;       // the ray this lane works for
  const uint32_t sub = lane & (G - 1u);     // its share of the spheres: sub, sub+G, ...
  const bool has_ray = ray < n_live_rays;
  const int src = __builtin_amdgcn_ds_bpermute(static_cast<int>(ray << 2), owner) << 2;
  f3 ro, rd;
  ro.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(o.x)));
  ro.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(o.y)));
  ro.z = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(o.z)));
  rd.x = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(d.x)));
  rd.y = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(d.y)));
  rd.z = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(d.z)));
  // lanes beyond the last ray work on a NaN origin: every test below fails for them (q and dn are NaN, "0 <= NaN" is
  // false, a NaN key is never below the best one), which spares the loop a select per sphere
  ro.x = has_ray ? ro.x : __uint_as_float(0x7fc00000u);
  uint32_t key = kNoHitKey;
  int slot = -1;
  const uint32_t n = n_sph_pad - kSphBatch;  // records that can hit (the tail records never do)
  RT_CNT(2, n >> shift)
  for (uint32_t base = 0; base < n; base += G) {
    const uint32_t j = base + sub;             // < n_sph_pad: n is a multiple of 8 and 8 more are readable
    const float4 g = sph[j];
    const float nx = ro.x - g.x, ny = ro.y - g.y, nz = ro.z - g.z;
    const float dn = fmaf(rd.z, nz, fmaf(rd.y, ny, rd.x * nx));
    const float q = fmaf(dn, dn, -(fmaf(nz, nz, fmaf(ny, ny, nx * nx)) - g.w));
    const unsigned long long cand = __builtin_amdgcn_fcmpf(fminf(q, -dn), 0.0f, 3 /* OGE */);
    if (cand != 0ull) {
      const uint32_t k = __float_as_uint(-dn - sqrt_for_hit(q)) - kEpsBits;
      const bool take = k < key;
      slot = take ? static_cast<int>(j) : slot;
      key = take ? k : key;
    }
  }
  // closest hit across the G lanes of a ray: lexicographic min of (key, slot)
  for (uint32_t off = 1; off < G; off <<= 1) {
    const uint32_t ok = static_cast<uint32_t>(__shfl_xor(static_cast<int>(key), static_cast<int>(off), 64));
    const int os = __shfl_xor(slot, static_cast<int>(off), 64);
    const bool take = (ok < key) | ((ok == key) & (os < slot));
    key = take ? ok : key;
    slot = take ? os : slot;
  }
  // back to the lane that owns the ray: the first lane of group `live_below`
  const int back = static_cast<int>((live_below << shift) << 2);
  SphereHit h;
  h.key = static_cast<uint32_t>(__builtin_amdgcn_ds_bpermute(back, static_cast<int>(key)));
  h.slot = __builtin_amdgcn_ds_bpermute(back, slot);
  return h;
}

// Sphere scan over a wave-uniform LIST of spheres (a bit mask per 64 spheres), ascending index.
// Used for the primary rays of a tile, whose mask comes from cull_tiles: a sphere outside the mask
// cannot be accepted by any ray of the tile, so skipping it does not change the result.
__device__ __forceinline__ SphereHit scan_spheres_listed(const float4 *__restrict__ sph, f3 o, f3 d,
                                                         const unsigned long long *__restrict__ col_mask,
                                                         const unsigned long long *__restrict__ row_mask, uint32_t words) {
  uint32_t best_key = kNoHitKey;
  int slot = -1;
  // Both words of a two-word mask (65..128 spheres: the reference's 80 slots) are requested before the first one is walked: read
  // inside the loop, the second word's loads would start when the first word's spheres are done — one more memory round trip per tile.
  unsigned long long m_next = (words > 1u) ? (col_mask[1] & row_mask[1]) : 0ull;
  for (uint32_t w = 0; w < words; ++w) {
    // a tile's spheres = those inside its column slab AND its row slab; the addresses are the same
    // in every lane: read once, keep in SGPRs
    unsigned long long m = (w == 0u) ? (col_mask[0] & row_mask[0]) : (w == 1u ? m_next : (col_mask[w] & row_mask[w]));
    m = (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<int>(m >> 32))) << 32) |
        static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(m)));
    RT_CNT(3, __popcll(m))
    while (m != 0ull) {
      const uint32_t i = w * 64u + static_cast<uint32_t>(__builtin_ctzll(m));
      m &= m - 1ull;
      const float4 g = sph[i];
      const float nx = o.x - g.x, ny = o.y - g.y, nz = o.z - g.z;
      const float dn = fmaf(d.z, nz, fmaf(d.y, ny, d.x * nx));
      const float q = fmaf(dn, dn, -(fmaf(nz, nz, fmaf(ny, ny, nx * nx)) - g.w));
      const unsigned long long cand = __builtin_amdgcn_fcmpf(fminf(q, -dn), 0.0f, 3 /* OGE */);
      if (cand != 0ull) {
        const uint32_t key = __float_as_uint(-dn - sqrt_for_hit(q)) - kEpsBits;
        slot = (key < best_key) ? static_cast<int>(i) : slot;
        best_key = min(best_key, key);
      }
    }
  }
  return SphereHit{best_key, slot};
}

// Inclusive prefix sum over the 64 lanes of a wave (DPP row shifts, then the two cross-row broadcasts of gfx9).
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t x) {
  x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x111, 0xf, 0xf, true));  // row_shr:1
  x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x112, 0xf, 0xf, true));  // row_shr:2
  x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x114, 0xf, 0xf, true));  // row_shr:4
  x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x118, 0xf, 0xf, true));  // row_shr:8
  x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x142, 0xa, 0xf, true));  // row_bcast:15 -> rows 1, 3
  x += static_cast<uint32_t>(__builtin_amdgcn_update_dpp(0, static_cast<int>(x), 0x143, 0xc, 0xf, true));  // row_bcast:31 -> rows 2, 3
  return x;
}

// The same scan for the kernels of large scenes (nothing staged: a sphere record is a memory round trip, and a thousand spheres are
// 16 mask words): the wave reads its mask words side by side — lane l the word l — compacts the set bits into a candidate list
// (ascending sphere index), fetches 64 candidates' records with ONE gather, and every ray then tests them from registers
// (v_readlane broadcasts): two dependent memory round trips per tile instead of one per mask word and one per candidate.
// Every lane of the wave must call it; `alive` = the lane carries a ray.  `scratch_base`: kListScratchBytes per wave.
__device__ __forceinline__ SphereHit scan_spheres_listed_wide(const float4 *__restrict__ sph, f3 o, f3 d, bool alive,
                                                              const unsigned long long *__restrict__ col_mask,
                                                              const unsigned long long *__restrict__ row_mask, uint32_t words,
                                                              unsigned char *scratch_base) {
  uint32_t tid = threadIdx.x;
  asm volatile("; thread index re-read per scan" : "+v"(tid));
  const uint32_t lane = tid & 63u;
  uint16_t *cand = reinterpret_cast<uint16_t *>(scratch_base + (tid >> 6) * kListScratchBytes + 64u * 64u);  // the pair list's space
  uint32_t best_key = kNoHitKey;
  int slot = -1;
  for (uint32_t wbase = 0u; wbase < words; wbase += 64u) {
    const uint32_t w = wbase + lane;
    unsigned long long m = 0ull;
    if (w < words) m = col_mask[w] & row_mask[w];
    const uint32_t cnt = static_cast<uint32_t>(__popcll(m));
    const uint32_t incl = wave_inclusive_sum(cnt);
    const uint32_t total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
    RT_CNT(3, total)
    for (uint32_t c0 = 0u; c0 < total; c0 += 64u) {  // 64 candidates at a time, ascending sphere index
      {
        uint32_t pos = incl - cnt - c0;  // place of this lane's first candidate in the window (wraps below it)
        unsigned long long mm = m;
        while (mm != 0ull) {
          const uint32_t bit = static_cast<uint32_t>(__builtin_ctzll(mm));
          mm &= mm - 1ull;
          if (pos < 64u) cand[pos] = static_cast<uint16_t>(w * 64u + bit);
          pos += 1u;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const uint32_t n_win = min(64u, total - c0);
      uint32_t idx = 0u;
      float4 g = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (lane < n_win) {
        idx = cand[lane];
        g = sph[idx];
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");  // the next window's writes stay behind these reads
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      for (uint32_t j = 0u; j < n_win; ++j) {
        const float gx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.x), static_cast<int>(j)));
        const float gy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.y), static_cast<int>(j)));
        const float gz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.z), static_cast<int>(j)));
        const float gw = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(g.w), static_cast<int>(j)));
        const int i = __builtin_amdgcn_readlane(static_cast<int>(idx), static_cast<int>(j));
        if (alive) {
          const float nx = o.x - gx, ny = o.y - gy, nz = o.z - gz;
          const float dn = fmaf(d.z, nz, fmaf(d.y, ny, d.x * nx));
          const float q = fmaf(dn, dn, -(fmaf(nz, nz, fmaf(ny, ny, nx * nx)) - gw));
          const unsigned long long hit = __builtin_amdgcn_fcmpf(fminf(q, -dn), 0.0f, 3 /* OGE */);
          if (hit != 0ull) {
            const uint32_t key = __float_as_uint(-dn - sqrt_for_hit(q)) - kEpsBits;
            slot = (key < best_key) ? i : slot;
            best_key = min(best_key, key);
          }
        }
      }
    }
  }
  return SphereHit{best_key, slot};
}

// ---- direction masks (rt_dirmask.h): secondary rays of scenes with up to 64 spheres --------------------------------
// The table entry of this lane's ray.  Lanes without a ray, and rays that miss the box of the registered balls, get the
// index of the table's extra all-zero entry (the one past the last cell), so that every lane can load its set without a
// branch around the load.  Returns false — wave-uniform — when some live lane's ray is not served by the table (origin
// outside the near box, direction not a unit vector, NaN): the wave then scans linearly as before.
__device__ __forceinline__ bool dirmask_lookup(const DirMaskParams &dm, f3 o, f3 d, bool alive, uint32_t &entry) {
  // unit direction (also false for a NaN anywhere in d); the origin's boxes are tested inner one first: a wave whose rays all
  // start inside the spheres' box — nearly every wave — needs neither the near-box test nor the entry point
  const bool unit = fabsf(dot3(d, d) - 1.0f) <= 4e-6f;
  const bool inside = (o.x >= dm.bmin[0]) & (o.x <= dm.bmax[0]) & (o.y >= dm.bmin[1]) & (o.y <= dm.bmax[1]) & (o.z >= dm.bmin[2]) &
                      (o.z <= dm.bmax[2]);  // false for a NaN in o
  if (__builtin_amdgcn_ballot_w64(alive & !unit) != 0ull) return false;
  f3 p = o;
  bool reaches = true;
  if (__builtin_amdgcn_ballot_w64(alive & !inside) != 0ull) {
    const bool near = (o.x >= dm.near_min[0]) & (o.x <= dm.near_max[0]) & (o.y >= dm.near_min[1]) & (o.y <= dm.near_max[1]) &
                      (o.z >= dm.near_min[2]) & (o.z <= dm.near_max[2]);  // the spheres' box lies inside the near box
    if (__builtin_amdgcn_ballot_w64(alive & !near) != 0ull) return false;
    // rays that start outside the box continue from the point where they enter it; a ray that misses it meets no sphere
    const bool outside = !inside;
    const float big = 1e30f;
    const float ivx = fabsf(d.x) > 1e-30f ? __builtin_amdgcn_rcpf(d.x) : (d.x < 0.0f ? -big : big);
    const float ivy = fabsf(d.y) > 1e-30f ? __builtin_amdgcn_rcpf(d.y) : (d.y < 0.0f ? -big : big);
    const float ivz = fabsf(d.z) > 1e-30f ? __builtin_amdgcn_rcpf(d.z) : (d.z < 0.0f ? -big : big);
    const float ax = (dm.bmin[0] - o.x) * ivx, bx = (dm.bmax[0] - o.x) * ivx;
    const float ay = (dm.bmin[1] - o.y) * ivy, by = (dm.bmax[1] - o.y) * ivy;
    const float az = (dm.bmin[2] - o.z) * ivz, bz = (dm.bmax[2] - o.z) * ivz;
    const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    reaches = !outside | (t_in <= t_out);
    p = outside ? fma3(d, t_in, o) : o;
  }
  const uint32_t cell = dirmask_cell(dm, p);
  const float inv_major = __builtin_amdgcn_rcpf(fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z)));
  const uint32_t bin = dirmask_bin(d, inv_major, dm.k, dm.half_k);
  // two more entries behind the table: the all-zero one for lanes without a ray, and for rays that miss the box the set of the spheres
  // far from all the others ("Outliers", rt_dirmask.h: every entry of the table holds them too) — empty for most scenes
  entry = alive ? (reaches ? cell * dm.bins + bin : dm.entries + 1u) : dm.entries;
  return true;
}

// The sphere part of ray.hlsl:97-107 over this lane's candidate set, lanes diverge: ascending slots, the arithmetic of
// the linear scan, strict '<' — the lexicographic (key, slot) minimum over the set, which is the linear scan's result
// because the set holds every sphere the scan could accept (rt_dirmask.h).
template <int WORDS32>
__device__ __forceinline__ SphereHit scan_spheres_masked(const float4 *__restrict__ sph, f3 o, f3 d, uint32_t m0, uint32_t m1,
                                                         uint32_t m2 = 0u, uint32_t m3 = 0u) {
  uint32_t best_key = kNoHitKey;
  int slot = -1;
  RT_CNT(4, 1)
  RT_CNT_LANE(6, __popc(m0) + __popc(m1) + __popc(m2) + __popc(m3))
#define RT_MASK_TEST(g, i)                                                                         \
  {                                                                                                \
    const float nx = o.x - g.x, ny = o.y - g.y, nz = o.z - g.z;                                    \
    const float dn = fmaf(d.z, nz, fmaf(d.y, ny, d.x * nx));                                       \
    const float q = fmaf(dn, dn, -(fmaf(nz, nz, fmaf(ny, ny, nx * nx)) - g.w));                    \
    if (fminf(q, -dn) >= 0.0f) {                                                                   \
      const uint32_t key = __float_as_uint(-dn - sqrt_for_hit(q)) - kEpsBits;                      \
      slot = (key < best_key) ? static_cast<int>(i) : slot;                                        \
      best_key = min(best_key, key);                                                               \
    }                                                                                              \
  }
  // One sphere per trip, two trips per loop turn with the records in alternating registers: the NEXT candidate's record
  // is requested before this one is tested, so that the LDS round trip hides behind the test instead of heading every
  // trip.  After the last candidate of a word the read goes to slot base + 32 (ctz of an empty word taken as 32): always
  // inside the staged image (it holds at least 3 records per sphere), and its result is dropped.
#if !RT_MASK_PIPELINE
#define RT_MASK_WORD(word_, base)                                                                  \
  while (word_ != 0u) {                                                                            \
    const uint32_t i = (base) + static_cast<uint32_t>(__builtin_ctz(word_));                       \
    word_ &= word_ - 1u;                                                                           \
    const float4 g = sph[i];                                                                       \
    RT_MASK_TEST(g, i)                                                                             \
  }
#else
#define RT_MASK_WORD(word_, base)                                                                  \
  if (__builtin_amdgcn_ballot_w64(word_ != 0u) != 0ull) {                                          \
    uint32_t ia = (base) + static_cast<uint32_t>(__builtin_ctzg(word_, 32)), ib;                   \
    float4 ga = sph[ia], gb;                                                                       \
    while (word_ != 0u) {                                                                          \
      RT_CNT_ANY(5, 1)                                                                             \
      word_ &= word_ - 1u;                                                                         \
      ib = (base) + static_cast<uint32_t>(__builtin_ctzg(word_, 32));                              \
      gb = sph[ib];                                                                                \
      RT_MASK_TEST(ga, ia)                                                                         \
      if (word_ == 0u) break;                                                                      \
      RT_CNT_ANY(5, 1)                                                                             \
      word_ &= word_ - 1u;                                                                         \
      ia = (base) + static_cast<uint32_t>(__builtin_ctzg(word_, 32));                              \
      ga = sph[ia];                                                                                \
      RT_MASK_TEST(gb, ib)                                                                         \
    }                                                                                              \
  }
#endif
  RT_MASK_WORD(m0, 0u)
  RT_MASK_WORD(m1, 32u)
  if (WORDS32 > 2) {  // scenes of 65..128 spheres
    RT_MASK_WORD(m2, 64u)
    RT_MASK_WORD(m3, 96u)
  }
#undef RT_MASK_WORD
#undef RT_MASK_TEST
  return SphereHit{best_key, slot};
}

// The same result as scan_spheres_masked for every lane, with the work dealt evenly: a lane's own walk costs the wave as
// many trips as its heaviest lane has candidates (8.8 on the headline scene, where the mean is 2.9).  Here every lane
// appends its (lane, slot) pairs to a list in LDS, the pairs are then tested 64 at a time by whichever lane they fall to
// (the ray is read from LDS), and a pair that hits does an LDS atomic min of (key, slot) on the ray's slot — the
// lexicographic minimum, which is what the strict '<' in ascending slot order selects.  Every lane of the wave must call it
// (lanes without a ray pass empty words).  `scratch_base`: the workgroup's scan scratch, kScanScratchBytes per wave.  Returns false
// (wave-uniform) when the wave has more pairs than the list holds: the caller walks the sets lane by lane.
template <int WORDS32>
__device__ __forceinline__ bool scan_spheres_balanced(const float4 *__restrict__ sph, f3 o, f3 d, uint32_t m0, uint32_t m1, uint32_t m2,
                                                      uint32_t m3, unsigned char *scratch_base, SphereHit &out) {
  // the thread index goes through an opaque move: everything derived from it (lane, the wave's scratch, the LDS addresses) is
  // then recomputed here, a few instructions per scan, instead of being kept in registers across the whole bounce loop —
  // at 72 registers the compiler otherwise spills exactly those values and reloads them from scratch memory in every scan
  uint32_t tid = threadIdx.x;
  asm volatile("; thread index re-read per scan" : "+v"(tid));
  const uint32_t lane = tid & 63u;
  unsigned char *scratch = scratch_base + (tid >> 6) * kScanScratchBytes;
  uint32_t p = static_cast<uint32_t>(__popc(m0) + __popc(m1));
  if (WORDS32 > 2) p += static_cast<uint32_t>(__popc(m2) + __popc(m3));
  const uint32_t incl = wave_inclusive_sum(p);
  const uint32_t total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
  if (total > kPairCap) return false;
  RT_CNT(4, 1)
  RT_CNT(5, (total + 63u) >> 6)
  RT_CNT_LANE(6, p)
  out = SphereHit{kNoHitKey, -1};
  if (total == 0u) return true;
  float4 *rays_a = reinterpret_cast<float4 *>(scratch);
  float2 *rays_b = reinterpret_cast<float2 *>(scratch + 64u * 16u);
  unsigned long long *best = reinterpret_cast<unsigned long long *>(scratch + 64u * 24u);
  uint16_t *pairs = reinterpret_cast<uint16_t *>(scratch + 64u * 32u);
  rays_a[lane] = make_float4(o.x, o.y, o.z, d.x);
  rays_b[lane] = make_float2(d.y, d.z);
  best[lane] = (static_cast<unsigned long long>(kNoHitKey) << 32) | 0xffffffffull;
  uint32_t n = incl - p;
  const uint32_t tag = lane << 7;
#define RT_PAIR_WORD(word_, base)                                                                  \
  while (word_ != 0u) {                                                                            \
    pairs[n] = static_cast<uint16_t>(tag | ((base) + static_cast<uint32_t>(__builtin_ctz(word_)))); \
    word_ &= word_ - 1u;                                                                           \
    n += 1u;                                                                                       \
  }
  RT_PAIR_WORD(m0, 0u)
  RT_PAIR_WORD(m1, 32u)
  if (WORDS32 > 2) {
    RT_PAIR_WORD(m2, 64u)
    RT_PAIR_WORD(m3, 96u)
  }
#undef RT_PAIR_WORD
  // LDS operations of one wave are carried out in the order they were issued; the fence only keeps the compiler from moving
  // the reads below above the writes of other lanes
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (uint32_t first = 0u; first < total; first += 64u) {  // wave-uniform trip count
    const uint32_t j = first + lane;
    if (j < total) {
      const uint32_t pr = pairs[j];
      const uint32_t owner = pr >> 7, i = pr & 127u;
      const float4 ra = rays_a[owner];
      const float2 rb = rays_b[owner];
      const float4 g = sph[i];
      const float nx = ra.x - g.x, ny = ra.y - g.y, nz = ra.z - g.z;
      const float dn = fmaf(rb.y, nz, fmaf(rb.x, ny, ra.w * nx));
      const float q = fmaf(dn, dn, -(fmaf(nz, nz, fmaf(ny, ny, nx * nx)) - g.w));
      if (fminf(q, -dn) >= 0.0f) {
        const uint32_t key = __float_as_uint(-dn - sqrt_for_hit(q)) - kEpsBits;
        // strictly below the initial distance, as the linear scan's '<' (ray.hlsl:98): a key of exactly kNoHitKey is no hit
        if (key < kNoHitKey) atomicMin(&best[owner], (static_cast<unsigned long long>(key) << 32) | i);
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const unsigned long long res = best[lane];
  out.key = static_cast<uint32_t>(res >> 32);
  out.slot = static_cast<int>(static_cast<uint32_t>(res));  // 0xffffffff = -1: nothing hit
  return true;
}

// ---- uniform grid walk (large scenes) -----------------------------------------------------------------------
// The sphere part of ray.hlsl:97-107 for ONE ray per lane, lanes diverge: a 3-D DDA over the grid of rt_grid.h,
// testing the spheres registered in every crossed cell with the arithmetic of the linear scan, until the best hit
// lies before the exit of the current cell (minus a margin) or the ray leaves the grid.
// Same result as the linear scan: (1) the per-sphere arithmetic is the same, so a sphere gives the same key
// wherever it is tested (a sphere registered in several cells is simply tested again); (2) the winner is the
// lexicographic minimum of (key, slot) — the linear scan's strict '<' in ascending slot order; (3) every sphere
// the float test can accept at distance t has the point o + t d inside its registered ball (rt_grid.h), hence in a
// cell the walk has visited once its cell-exit parameter has passed t; the walk only stops at a hit that is
// t_margin before the current exit, so no unvisited sphere can have an equal or smaller distance.  The
// traversal's own arithmetic (approximate reciprocals, accumulated boundary parameters) only moves cell
// boundaries by far less than the registration slack.  RT_OPT_CULL = 2 re-scans linearly and counts disagreements.
__device__ __forceinline__ SphereHit scan_grid(const SceneView &sv, const GridParams &gp, f3 o, f3 d) {
  uint32_t best_key = kNoHitKey;
  int best_slot = -1;
  const float big = 1e30f;
  const float ivx = fabsf(d.x) > 1e-30f ? __builtin_amdgcn_rcpf(d.x) : (d.x < 0.0f ? -big : big);
  const float ivy = fabsf(d.y) > 1e-30f ? __builtin_amdgcn_rcpf(d.y) : (d.y < 0.0f ? -big : big);
  const float ivz = fabsf(d.z) > 1e-30f ? __builtin_amdgcn_rcpf(d.z) : (d.z < 0.0f ? -big : big);
  // ray against the grid's box
  const float ax = (gp.gmin[0] - o.x) * ivx, bx = (gp.gmax[0] - o.x) * ivx;
  const float ay = (gp.gmin[1] - o.y) * ivy, by = (gp.gmax[1] - o.y) * ivy;
  const float az = (gp.gmin[2] - o.z) * ivz, bz = (gp.gmax[2] - o.z) * ivz;
  const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
  const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
  if (!(t_in <= t_out)) return SphereHit{best_key, best_slot};  // misses the box (or NaN): no sphere can be accepted
  // the cell of the entry point
  const f3 p = fma3(d, t_in, o);
  const int nx = static_cast<int>(gp.nx), ny = static_cast<int>(gp.ny), nz = static_cast<int>(gp.nz);
  int ix = min(max(static_cast<int>(floorf((p.x - gp.gmin[0]) * gp.inv_cell[0])), 0), nx - 1);
  int iy = min(max(static_cast<int>(floorf((p.y - gp.gmin[1]) * gp.inv_cell[1])), 0), ny - 1);
  int iz = min(max(static_cast<int>(floorf((p.z - gp.gmin[2]) * gp.inv_cell[2])), 0), nz - 1);
  const int sx = d.x >= 0.0f ? 1 : -1, sy = d.y >= 0.0f ? 1 : -1, sz = d.z >= 0.0f ? 1 : -1;
  // parameter at which the ray leaves the current cell along each axis, and its increment per cell
  float tx = fmaxf((fmaf(static_cast<float>(ix + (sx > 0 ? 1 : 0)), gp.cell[0], gp.gmin[0]) - o.x) * ivx, t_in);
  float ty = fmaxf((fmaf(static_cast<float>(iy + (sy > 0 ? 1 : 0)), gp.cell[1], gp.gmin[1]) - o.y) * ivy, t_in);
  float tz = fmaxf((fmaf(static_cast<float>(iz + (sz > 0 ? 1 : 0)), gp.cell[2], gp.gmin[2]) - o.z) * ivz, t_in);
  const float dtx = gp.cell[0] * fabsf(ivx), dty = gp.cell[1] * fabsf(ivy), dtz = gp.cell[2] * fabsf(ivz);
  for (uint32_t guard = 0; guard < 3u * kGridMaxDimDevice + 3u; ++guard) {  // a walk crosses at most nx + ny + nz cells
    const uint32_t cell = static_cast<uint32_t>((iz * ny + iy) * nx + ix);
    const uint32_t first = sv.grid_cells[cell], last = sv.grid_cells[cell + 1u];
    for (uint32_t k = first; k < last; ++k) {
      const int slot = static_cast<int>(sv.grid_items[k]);
      const float4 g = sv.sph[slot];
      const float nxv = o.x - g.x, nyv = o.y - g.y, nzv = o.z - g.z;
      const float dn = fmaf(d.z, nzv, fmaf(d.y, nyv, d.x * nxv));
      const float q = fmaf(dn, dn, -(fmaf(nzv, nzv, fmaf(nyv, nyv, nxv * nxv)) - g.w));
      if (fminf(q, -dn) >= 0.0f) {
        const uint32_t key = __float_as_uint(-dn - sqrt_for_hit(q)) - kEpsBits;
        const bool take = (key < best_key) | ((key == best_key) & (slot < best_slot));
        best_slot = take ? slot : best_slot;
        best_key = take ? key : best_key;
      }
    }
    const float t_exit = fminf(tx, fminf(ty, tz));
    if (best_key != kNoHitKey && __uint_as_float(best_key + kEpsBits) < t_exit - gp.t_margin) break;
    if (tx <= ty && tx <= tz) {
      ix += sx;
      tx += dtx;
      if (ix < 0 || ix >= nx) break;
    } else if (ty <= tz) {
      iy += sy;
      ty += dty;
      if (iy < 0 || iy >= ny) break;
    } else {
      iz += sz;
      tz += dtz;
      if (iz < 0 || iz >= nz) break;
    }
  }
  return SphereHit{best_key, best_slot};
}

// A 16-byte record of the scene image read at a wave-uniform address.  CONST = the record lives in device memory (the kernels of
// large scenes stage nothing): it is read through the constant address space — the image is never written by a kernel —, which
// makes it a SCALAR load: it does not queue up behind the wave's vector loads (a vector load here would wait, at its
// s_waitcnt vmcnt(0), for the candidate-list entry requested just before it), and it returns into scalar registers.
template <bool CONST>
__device__ __forceinline__ float4 load_record(const float4 *p) {
  if (CONST) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const __attribute__((address_space(4))) f32x4 *q = (const __attribute__((address_space(4))) f32x4 *)(p);
    const f32x4 v = *q;
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *p;
}

// The rest of ray.hlsl:97-107: sphere result -> Hit, then the planes (ray.hlsl:69-94).
template <bool CONST = false>
__device__ __forceinline__ Hit finish_hit(const SceneView &sv, SphereHit sh, f3 o, f3 d) {
  const float best = __uint_as_float(sh.key + kEpsBits);
  const int slot = sh.slot;
  Hit h;
  h.t = best;
  h.slot = slot;
  h.plane = false;
  h.obj = (slot >= 0) ? sv.sph_obj[slot] : -1;
  for (uint32_t j = 0; j < sv.n_pl; ++j) {
    const float4 a = load_record<CONST>(sv.pl_a + j), b = load_record<CONST>(sv.pl_b + j);
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

// The plane part of ray.hlsl:97-107 on its own (ray.hlsl:69-94), and its merge with the sphere result: the same
// lexicographic (distance, object index) minimum finish_hit builds, in another order — so that the planes can be tested
// while the ray's candidate set is still on its way from memory (dirmask_lookup).
template <bool CONST = false>
__device__ __forceinline__ Hit nearest_plane(const SceneView &sv, f3 o, f3 d) {
  Hit h;
  h.t = 1000000.0f;  // ray.hlsl:98: the initial hit.distance
  h.obj = -1;
  h.slot = -1;
  h.plane = false;
  for (uint32_t j = 0; j < sv.n_pl; ++j) {
    const float4 a = load_record<CONST>(sv.pl_a + j), b = load_record<CONST>(sv.pl_b + j);
    const f3 c = mk3(a.x, a.y, a.z), nrm = mk3(b.x, b.y, b.z);
    const int idx = static_cast<int>(f2u(b.w));
    const float den = dot3(d, nrm);
    if (fabsf(den) < kEpsilon) continue;
    const float num = dot3(c - o, nrm);
    const float t = num / den;
    const f3 p = fma3(d, t, o);
    const bool outside = (fabsf(p.x - c.x) > a.w) || (fabsf(p.y - c.y) > a.w) || (fabsf(p.z - c.z) > a.w);
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
__device__ __forceinline__ Hit merge_sphere_hit(const SceneView &sv, Hit h, SphereHit sh) {
  if (sh.slot >= 0) {
    const float t = __uint_as_float(sh.key + kEpsBits);
    const int obj = sv.sph_obj[sh.slot];
    // an equal distance only wins with the lower object index (planes and spheres share one index space)
    if ((t < h.t) | ((t == h.t) & ((h.obj < 0) | (obj < h.obj)))) {
      h.t = t;
      h.obj = obj;
      h.slot = sh.slot;
      h.plane = false;
    }
  }
  return h;
}

// The same for kernels that never report object indices (sample-only launches) on scenes whose index table is a memory round trip
// away: the sphere's object index is only fetched when its distance EQUALS the plane hit's (the tie-break of ray.hlsl:97-107's
// ascending order); otherwise any non-negative value says "hit".
__device__ __forceinline__ Hit merge_sphere_hit_lazy(const SceneView &sv, Hit h, SphereHit sh) {
  if (sh.slot >= 0) {
    const float t = __uint_as_float(sh.key + kEpsBits);
    int obj = 0;
    bool wins = t < h.t;
    if (t == h.t) {  // rare
      obj = sv.sph_obj[sh.slot];
      wins = (h.obj < 0) | (obj < h.obj);
    }
    if (wins) {
      h.t = t;
      h.obj = obj;
      h.slot = sh.slot;
      h.plane = false;
    }
  }
  return h;
}

// ray_intersect (ray.hlsl:97-107) for the lanes of a wave whose rays the direction-mask table serves: the set is loaded
// (every lane, no branch: lanes without a ray read the table's zero entry), the planes are tested while it is on its way,
// then each lane tests its candidates.  One instance per entry width, so that the loaded registers go straight into the
// loop (a merge of the two widths would force the wait for the load to the spot of the load).
template <bool LEAN>
__device__ __forceinline__ Hit intersect_masked(const SceneView &sv, const FrameParams &fp, f3 o, f3 d, bool alive, uint32_t entry,
                                                Hit miss, unsigned char *scratch = nullptr) {
  const DirMaskParams &dm = *sv.dm;
  Hit hp = miss;
  SphereHit sh;
#ifdef RT_PROFILE_EVENTS
  const int n_rays_ = __popcll(__builtin_amdgcn_ballot_w64(alive));  // every lane takes part in the ballot; lane 0 adds the count
  RT_CNT(7, n_rays_)
#endif
  // RT_PLANES_AFTER_SCAN: the planes are tested after the balanced scan instead of under the table load's latency — four
  // registers fewer are live through the scan, which is what lets the multi-frame kernel keep 7 waves per SIMD
  const bool planes_late = RT_PLANES_AFTER_SCAN && RT_BALANCED_SCAN && scratch != nullptr;
  if (dm.words == 1u) {  // wave-uniform
    const uint2 w = *reinterpret_cast<const uint2 *>(dm.table + entry);
    if (!planes_late && alive) hp = nearest_plane(sv, o, d);
    if (!(RT_BALANCED_SCAN && scratch != nullptr && scan_spheres_balanced<2>(sv.sph, o, d, w.x, w.y, 0u, 0u, scratch, sh)))
      sh = scan_spheres_masked<2>(sv.sph, o, d, w.x, w.y);
  } else {
    const uint4 w = *reinterpret_cast<const uint4 *>(dm.table + 2u * static_cast<size_t>(entry));
    if (!planes_late && alive) hp = nearest_plane(sv, o, d);
    if (!(RT_BALANCED_SCAN && scratch != nullptr && scan_spheres_balanced<4>(sv.sph, o, d, w.x, w.y, w.z, w.w, scratch, sh)))
      sh = scan_spheres_masked<4>(sv.sph, o, d, w.x, w.y, w.z, w.w);
  }
  if (planes_late && alive) hp = nearest_plane(sv, o, d);
  if (!LEAN && fp.cull_check && alive) {  // self-check frame: the linear scan must agree
    const SphereHit full = scan_spheres(sv.sph, 0u, sv.n_sph_pad - kSphBatch, o, d);
    if (full.key != sh.key || full.slot != sh.slot) atomicAdd(&fp.counters[3], 1ull);
  }
  return alive ? merge_sphere_hit(sv, hp, sh) : miss;
}

// ---- candidate lists (rt_dirmask.h "Candidate lists"): secondary rays of scenes with more than 128 spheres ------------
// ray_intersect's sphere part for a whole wave (every lane calls it; `alive` = the lane carries a ray) in ROUNDS: every
// ray that is not done looks the entry of the point it has reached up (32 bytes: up to 15 sphere slots and the entry's
// reach), the wave tests all its (ray, sphere) pairs 64 at a time — whichever lane a pair falls to reads ray and slot from
// LDS, the sphere record from memory, and a pair that hits does an LDS atomic min of (key << 32 | slot) on its ray's word:
// the lexicographic (key, slot) minimum, i.e. the linear scan's strict '<' in ascending slot order — and a ray whose
// best hit (sphere or plane) lies inside the reach is done; the others advance and look up again.  Rays the table does
// not serve (direction not a unit vector, NaN, farther along than s_cap, an entry without a reach) scan linearly at the end,
// together (the cooperative scan when they are few).  The planes are tested under the first round's load.
// `scratch_base`: kListScratchBytes per wave.
#ifndef RT_LIST_OUTLIERS
#define RT_LIST_OUTLIERS 1
#endif
template <bool LEAN, bool TRIPS2 = (RT_LIST_TRIPS2 != 0)>
__device__ __forceinline__ Hit intersect_listed(const SceneView &sv, const FrameParams &fp, f3 o, f3 d, bool alive, Hit miss,
                                                unsigned char *scratch_base RT_PH_PARAMS) {
  const DirMaskParams &dm = *sv.dm;
  uint32_t tid = threadIdx.x;
  asm volatile("; thread index re-read per scan" : "+v"(tid));
  const uint32_t lane = tid & 63u;
  unsigned char *scratch = scratch_base + (tid >> 6) * kListScratchBytes;
  float4 *rays_a = reinterpret_cast<float4 *>(scratch);
  float2 *rays_b = reinterpret_cast<float2 *>(scratch + 64u * 16u);
  unsigned long long *best = reinterpret_cast<unsigned long long *>(scratch + 64u * 24u);
  uint4 *raw = reinterpret_cast<uint4 *>(scratch + 64u * 32u);
  const uint16_t *raw16 = reinterpret_cast<const uint16_t *>(raw);
  uint16_t *pairs = reinterpret_cast<uint16_t *>(scratch + 64u * 64u);
  RT_CNT(4, 1)
#ifdef RT_PROFILE_EVENTS
  const int n_rays_ = __popcll(__builtin_amdgcn_ballot_w64(alive));
  RT_CNT(7, n_rays_)
#endif
  const bool unit = fabsf(dot3(d, d) - 1.0f) <= 4e-6f;  // false for a NaN in d
  const bool inside = (o.x >= dm.bmin[0]) & (o.x <= dm.bmax[0]) & (o.y >= dm.bmin[1]) & (o.y <= dm.bmax[1]) & (o.z >= dm.bmin[2]) &
                      (o.z <= dm.bmax[2]);  // false for a NaN in o
  bool todo = alive & unit;  // still looking entries up
  bool far = alive & !unit;  // scans linearly at the end
  float s = 0.0f;            // ray parameter of the next lookup
  if (__builtin_amdgcn_ballot_w64(todo & !inside) != 0ull) {
    // rays that start outside the box of the registered balls continue from the point where they enter it; a ray that
    // misses it meets no sphere.  A NaN origin fails every compare below: the ray ends up `far`.
    const float big = 1e30f;
    const float ivx = fabsf(d.x) > 1e-30f ? __builtin_amdgcn_rcpf(d.x) : (d.x < 0.0f ? -big : big);
    const float ivy = fabsf(d.y) > 1e-30f ? __builtin_amdgcn_rcpf(d.y) : (d.y < 0.0f ? -big : big);
    const float ivz = fabsf(d.z) > 1e-30f ? __builtin_amdgcn_rcpf(d.z) : (d.z < 0.0f ? -big : big);
    const float ax = (dm.bmin[0] - o.x) * ivx, bx = (dm.bmax[0] - o.x) * ivx;
    const float ay = (dm.bmin[1] - o.y) * ivy, by = (dm.bmax[1] - o.y) * ivy;
    const float az = (dm.bmin[2] - o.z) * ivz, bz = (dm.bmax[2] - o.z) * ivz;
    const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    const bool outside = todo & !inside;
    const bool finite = (o.x == o.x) & (o.y == o.y) & (o.z == o.z);
    const bool misses = outside & finite & !(t_in <= t_out);
    const bool beyond = outside & (!finite | (t_in > dm.s_cap));  // (a NaN t_in with a finite origin cannot happen: d is a unit vector)
    // Spheres far from all the others ("Outliers", rt_dirmask.h) are listed by EVERY entry: with such a scene a ray that misses the box
    // still looks one entry up, where it starts (the cell index clamps into the grid; that cell's other spheres are tested for nothing),
    // and leaves the rounds at its first step.
    const bool keep_missing = RT_LIST_OUTLIERS && dm.n_outliers != 0u;
    s = (outside & !(misses & keep_missing)) ? t_in : 0.0f;
    far |= beyond;
    todo &= !(misses & !keep_missing) & !beyond;
  }
  const float inv_major = __builtin_amdgcn_rcpf(fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z)));
  const uint32_t bin = dirmask_bin(d, inv_major, dm.k, dm.half_k);
  rays_a[lane] = make_float4(o.x, o.y, o.z, d.x);
  rays_b[lane] = make_float2(d.y, d.z);
  best[lane] = (static_cast<unsigned long long>(kNoHitKey) << 32) | 0xffffffffull;
  Hit hp = miss;
  bool planes_pending = true;
  const uint4 *table = reinterpret_cast<const uint4 *>(dm.table);
  RT_PHL(0)
  for (uint32_t round = 0u; __builtin_amdgcn_ballot_w64(todo) != 0ull; ++round) {
    if (round == 48u) {  // rays that crawl through entries of tiny reaches (heaps of overlapping balls): the linear scan serves them
      far |= todo;
      break;
    }
    const f3 p = fma3(d, s, o);
    const uint32_t entry = todo ? dirmask_cell(dm, p) * dm.bins + bin : dm.entries;  // lanes without a lookup read the zero entry
#if RT_LIST_NT_LOADS
    // (streamed: an entry is read once — it should not push the sphere records and materials out of the caches)
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *tv = reinterpret_cast<const u32x4 *>(table) + 2u * static_cast<size_t>(entry);
    const u32x4 v0 = __builtin_nontemporal_load(tv), v1 = __builtin_nontemporal_load(tv + 1);
    const uint4 w0 = make_uint4(v0.x, v0.y, v0.z, v0.w), w1 = make_uint4(v1.x, v1.y, v1.z, v1.w);
#else
    const uint4 w0 = table[2u * static_cast<size_t>(entry)], w1 = table[2u * static_cast<size_t>(entry) + 1u];
#endif
    if (planes_pending) {  // wave-uniform: the first round
      if (alive) hp = nearest_plane<true>(sv, o, d);
      planes_pending = false;
    }
    RT_PHL_WAIT
    RT_PHL(1)
    raw[2u * lane] = w0;
    raw[2u * lane + 1u] = w1;
    const uint32_t n = todo ? (w0.x >> 12) & 15u : 0u;
    const uint32_t q = w0.x & 0xfffu;
    const uint32_t incl = wave_inclusive_sum(n);
    const uint32_t total = static_cast<uint32_t>(__builtin_amdgcn_readlane(static_cast<int>(incl), 63));
    RT_CNT(5, (total + 63u) >> 6)
    RT_CNT_LANE(6, n)
    RT_CNT(2, 1)  // rounds
    {
      uint16_t *mine = pairs + (incl - n);
      const uint32_t tag = lane << 8;
      for (uint32_t j = 0; j < n; ++j) mine[j] = static_cast<uint16_t>(tag | (j + 1u));
    }
    // LDS operations of one wave are carried out in the order they were issued; the fences only keep the compiler from moving
    // the reads below above the writes of other lanes
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    RT_PHL(2)
    if (TRIPS2) {
    // two pairs per lane and trip: the two chains of dependent reads (pair -> slot -> sphere record, the last one from memory) are
    // in flight together.  Lanes beyond the list read pair 0's operands (always there when total > 0) and drop the result.
    for (uint32_t first = 0u; first < total; first += 128u) {  // wave-uniform trip count
      const uint32_t ja = first + lane, jb = ja + 64u;
      const bool va = ja < total, vb = jb < total;
      const uint32_t pra = pairs[va ? ja : 0u], prb = pairs[vb ? jb : 0u];
      const uint32_t owner_a = pra >> 8, owner_b = prb >> 8;
      const uint32_t ia = raw16[owner_a * 16u + (pra & 255u)], ib = raw16[owner_b * 16u + (prb & 255u)];
      const float4 raa = rays_a[owner_a], rab = rays_a[owner_b];
      const float2 rba = rays_b[owner_a], rbb = rays_b[owner_b];
      const float4 ga = sv.sph[ia], gb = sv.sph[ib];
      {
        const float nx = raa.x - ga.x, ny = raa.y - ga.y, nz = raa.z - ga.z;
        const float dn = fmaf(rba.y, nz, fmaf(rba.x, ny, raa.w * nx));
        const float qd = fmaf(dn, dn, -(fmaf(nz, nz, fmaf(ny, ny, nx * nx)) - ga.w));
        if (va && fminf(qd, -dn) >= 0.0f) {
          const uint32_t key = __float_as_uint(-dn - sqrt_for_hit(qd)) - kEpsBits;
          if (key < kNoHitKey) atomicMin(&best[owner_a], (static_cast<unsigned long long>(key) << 32) | ia);
        }
      }
      if (__builtin_amdgcn_ballot_w64(vb) != 0ull) {
        const float nx = rab.x - gb.x, ny = rab.y - gb.y, nz = rab.z - gb.z;
        const float dn = fmaf(rbb.y, nz, fmaf(rbb.x, ny, rab.w * nx));
        const float qd = fmaf(dn, dn, -(fmaf(nz, nz, fmaf(ny, ny, nx * nx)) - gb.w));
        if (vb && fminf(qd, -dn) >= 0.0f) {
          const uint32_t key = __float_as_uint(-dn - sqrt_for_hit(qd)) - kEpsBits;
          if (key < kNoHitKey) atomicMin(&best[owner_b], (static_cast<unsigned long long>(key) << 32) | ib);
        }
      }
    }
    } else {
    for (uint32_t first = 0u; first < total; first += 64u) {  // wave-uniform trip count
      const uint32_t j = first + lane;
      if (j < total) {
        const uint32_t pr = pairs[j];
        const uint32_t owner = pr >> 8;
        const uint32_t i = raw16[owner * 16u + (pr & 255u)];
        const float4 ra = rays_a[owner];
        const float2 rb = rays_b[owner];
        const float4 g = sv.sph[i];
        const float nx = ra.x - g.x, ny = ra.y - g.y, nz = ra.z - g.z;
        const float dn = fmaf(rb.y, nz, fmaf(rb.x, ny, ra.w * nx));
        const float qd = fmaf(dn, dn, -(fmaf(nz, nz, fmaf(ny, ny, nx * nx)) - g.w));
        if (fminf(qd, -dn) >= 0.0f) {
          const uint32_t key = __float_as_uint(-dn - sqrt_for_hit(qd)) - kEpsBits;
          if (key < kNoHitKey) atomicMin(&best[owner], (static_cast<unsigned long long>(key) << 32) | i);
        }
      }
    }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    RT_PHL_WAIT
    RT_PHL(3)
    if (todo) {
      const uint32_t key = static_cast<uint32_t>(best[lane] >> 32);
      const float t_sphere = __uint_as_float(key + kEpsBits);  // 1e6 when nothing has hit: beyond any reach that matters
      const float t_near = hp.obj >= 0 ? fminf(hp.t, t_sphere) : t_sphere;
      const float reach = static_cast<float>(q) * dm.reach_unit;
      if (q == 0u) {  // the entry serves no ray (too many overlapping balls)
        far = true;
        todo = false;
      } else if (t_near <= fmaf(reach, kDirListDone, s)) {
        todo = false;  // every sphere not tested yet can only be accepted beyond s + reach
      } else {
        s = fmaf(reach, kDirListStep, s);
        const f3 pn = fma3(d, s, o);
        const bool in_box = (pn.x >= dm.bmin[0]) & (pn.x <= dm.bmax[0]) & (pn.y >= dm.bmin[1]) & (pn.y <= dm.bmax[1]) &
                            (pn.z >= dm.bmin[2]) & (pn.z <= dm.bmax[2]);
        if (!in_box) todo = false;  // left the box of the registered balls (it is convex: the ray does not come back)
        else if (s > dm.s_cap) {
          far = true;
          todo = false;
        }
      }
    }
  }
  RT_PHL(4)
  if (planes_pending && alive) hp = nearest_plane<true>(sv, o, d);  // no lane looked anything up
  SphereHit sh;
  {
    const unsigned long long res = best[lane];
    sh.key = static_cast<uint32_t>(res >> 32);
    sh.slot = static_cast<int>(static_cast<uint32_t>(res));  // 0xffffffff = -1: nothing hit
  }
  const unsigned long long far_mask = __builtin_amdgcn_ballot_w64(far);
  if (far_mask != 0ull) {  // rays the table does not serve: the linear scan (its result stands alone: it tests every sphere)
    asm volatile("; rare path: rays outside the candidate lists' domain" ::: "memory");
    const uint32_t n_far = static_cast<uint32_t>(__popcll(far_mask));
    SphereHit lin{kNoHitKey, -1};
    if (n_far <= 32u) lin = scan_spheres_coop(sv.sph, sv.n_sph_pad, o, d, far, far_mask, n_far);
    else if (far) lin = scan_spheres(sv.sph, 0u, sv.n_sph_pad - kSphBatch, o, d);
    if (far) sh = lin;
  }
  const Hit merged = alive ? ((LEAN && RT_LAZY_OBJ) ? merge_sphere_hit_lazy(sv, hp, sh) : merge_sphere_hit(sv, hp, sh)) : miss;
  if (!LEAN && fp.cull_check && alive) {
    // self-check frame: the linear scan must lead to the same hit.  (The sphere results themselves may differ: a plane hit inside
    // the reach ends the rounds, and spheres farther than that plane are then never tested — they could not win.)
    const SphereHit full = scan_spheres(sv.sph, 0u, sv.n_sph_pad - kSphBatch, o, d);
    const Hit want = merge_sphere_hit(sv, hp, full);
    if (want.t != merged.t || want.obj != merged.obj || want.slot != merged.slot || want.plane != merged.plane) atomicAdd(&fp.counters[3], 1ull);
  }
  RT_PHL_WAIT
  RT_PHL(5)
  return merged;
}

// State of one pixel's path between two iterations of frag.hlsl:36-43.
struct Path {
  f3 o, d, energy, result;
  Rng rng;
  uint32_t p;      // local pixel index (local_row * width + x)
  uint32_t seg;    // iterations executed so far
  uint32_t rough;  // of which drew random() (rough hit)
};

// fs_main up to the call of trace_ray_with_reflections (frag.hlsl:49-57, ray.hlsl:31-37).
// `frame`: which frame of the launch's batch (stacked frame_tile_rows tile rows apart in the outputs).
__device__ __forceinline__ void path_begin(const FrameParams &fp, uint32_t x, uint32_t lrow, uint32_t frame, Path &s) {
  // local row -> global row of the round-robin strip partition
  // (one GPU: the local rows are the frame's rows — spares the per-lane integer division)
  const uint32_t y = fp.world == 1u ? lrow : ((lrow / fp.strip_rows) * fp.world + fp.rank) * fp.strip_rows + lrow % fp.strip_rows;
  s.p = (frame * fp.frame_tile_rows * kTile + lrow) * fp.width + x;
  // D14: SV_POSITION.xy = pixel centre
  const float pjx = (static_cast<float>(x) + 0.5f) + fp.jitter[2u * frame];
  const float pjy = (static_cast<float>(y) + 0.5f) + fp.jitter[2u * frame + 1u];
  const float cjx = pjx / fp.width_f, cjy = pjy / fp.height_f;
  s.rng = Rng{42.0f, cjx, cjy};
  f3 pw = fma3(fp.cam_right, cjx * fp.vp_width, fp.top_left);
  pw = fma3(fp.cam_neg_up, cjy * fp.vp_height, pw);
  s.o = fp.cam_pos;
  s.d = normalize3(pw - fp.cam_pos);
  s.energy = mk3(1.0f, 1.0f, 1.0f);
  s.result = mk3(0.0f, 0.0f, 0.0f);
  s.seg = 0;
  s.rough = 0;
}

// shade() of one iteration of frag.hlsl:36-43 for the hit `h`; x1, x2 are the two random() texel values
// requested during the scan (used if the hit is rough), seed1 the RNG state between them.  Returns whether the
// loop goes on.
// DEFER_SKY: on a miss only mark the path; its skybox texel is fetched in path_end, together with the other
// finished lanes of the wave instead of a few lanes at a time after every bounce.
// LEAN: the launch writes float samples only (multi-frame launches and the single-frame launches in flight: no
// per-frame diagnostic outputs, no history, the sRGB store happens in resolve_batch_kernel) — the kernel is
// compiled without those paths.
// X2_GIVEN: x2 and seed1 are the second random() texel and the RNG state it was addressed with (requested by the caller, early);
// otherwise the texel is requested here, when the hit turns out rough.
template <bool DEFER_SKY, bool LEAN = false, bool X2_GIVEN = (RT_X2_PREFETCH != 0)>
__device__ __forceinline__ bool path_shade(const SceneView &sv, const FrameParams &fp, Path &s, const Hit h,
                                           const float x1, const float x2, const float seed1) {
  if (!LEAN && fp.hits) fp.hits[static_cast<size_t>(s.p) * fp.reflection_limit + s.seg] = h.obj;
  s.seg += 1;
  if (h.obj < 0) {  // frag.hlsl:14-17: miss -> skybox, energy 0, loop ends (|0| < EPSILON)
    if (DEFER_SKY) {
      s.rough |= 0x80000000u;  // "ended on a miss": path_end<true> adds energy * skybox(d)
    } else {
      s.result = fma3(s.energy, sample_hdri(fp, s.d), s.result);
    }
    return false;
  }
  const f3 pos = fma3(s.d, h.t, s.o);
  f3 nrm;
  float4 m0, m1;  // materials by slot: no object index needed first
  if (h.plane) {
    const float4 b = sv.pl_b[h.slot];
    nrm = mk3(b.x, b.y, b.z);  // ray.hlsl:43-45: stored normal, not renormalised
    m0 = sv.pmat0[h.slot];
    m1 = sv.pmat1[h.slot];
  } else {
    const float4 g = sv.sph[h.slot];
    m0 = sv.smat0[h.slot];
    m1 = sv.smat1[h.slot];
    nrm = normalize3(pos - mk3(g.x, g.y, g.z));
  }
  s.o = fma3(nrm, 0.001f, pos);                   // frag.hlsl:21
  const float k = 2.0f * dot3(nrm, s.d);          // reflect(), frag.hlsl:23
  f3 nd = fma3(nrm, -k, s.d);
  if (m0.w != 0.0f) {                             // random.hlsl:26-28
    if (X2_GIVEN) {
      nd = hemisphere(s.rng, nd, m1.w, x1, x2, seed1);
    } else {
      const float seed_a = rng_next_seed(s.rng.seed, x1);
      const float x2_now = fp.rnd[rng_texel(fp, Rng{seed_a, s.rng.px, s.rng.py})];
      nd = hemisphere(s.rng, nd, m1.w, x1, x2_now, seed_a);
    }
    s.rough += 1;
  }
  s.d = nd;
  const float cosine = saturate(dot3(nrm, s.d));
  s.result = fma3(s.energy, mk3(m1.x, m1.y, m1.z), s.result);  // D1: energy before the update
  s.energy = s.energy * (mk3(m0.x, m0.y, m0.z) * cosine);      // frag.hlsl:27
  // frag.hlsl:40-42 `length(energy) < EPSILON`: the correctly rounded root is monotonic, so the test is a
  // compare of the squared length with the smallest float whose root reaches EPSILON (0x2b8cbccb:
  // sqrtf(it) >= 1e-6f, sqrtf(its predecessor) < 1e-6f; tests/test_oracle_math.py) — no root needed.
  if (dot3(s.energy, s.energy) < __uint_as_float(0x2b8cbccbu)) return false;
  return s.seg < fp.reflection_limit;
}

// One iteration of frag.hlsl:36-43 (ray_intersect + shade) for a whole wave (every lane calls it; `alive` = this
// lane carries a ray).  The first random() texel is requested before the scan and used after it if the hit is
// rough: the gather's latency (an L2 round trip) hides under the object scan.  Picks the
// wave-cooperative scan when at most 32 lanes are live.  Returns the lane's new `alive`.
// MODE: how secondary rays find their spheres — 0 the direction-mask table of small scenes (or the linear / cooperative scan), 1 the
// walk through the sphere grid, 2 the candidate lists of large scenes (`scratch`: kListScratchBytes per wave).
template <bool DEFER_SKY, int MODE = 0, bool LEAN = false>
__device__ __forceinline__ bool path_step_wave(const SceneView &sv, const FrameParams &fp, Path &s, bool alive,
                                               const unsigned long long *__restrict__ primary_mask,
                                               const unsigned long long *__restrict__ primary_mask2 RT_PH_PARAMS,
                                               const GridParams *gp = nullptr, unsigned char *scratch = nullptr) {
  constexpr bool GRID = MODE == 1;
  const unsigned long long live = __builtin_amdgcn_ballot_w64(alive);
  const uint32_t n_live = static_cast<uint32_t>(__popcll(live));
  if (n_live == 0u) return false;
  // Both random() texels of a rough hit are requested during the scan, whether or not the hit turns out rough
  // (random.hlsl:4-11: the second texel's address needs the first one's value, nothing else): their L2 round
  // trips hide under the object scan instead of heading the shading chain one after the other.
  float x1 = 0.0f, x2 = 0.0f, seed1 = 0.0f;
  if (alive) x1 = fp.rnd[rng_texel(fp, s.rng)];
#if RT_X2_PREFETCH
#define RT_REQUEST_X2                                          \
  if (alive) {                                                 \
    seed1 = rng_next_seed(s.rng.seed, x1);                     \
    x2 = fp.rnd[rng_texel(fp, Rng{seed1, s.rng.px, s.rng.py})]; \
  }
#else
#define RT_REQUEST_X2
#endif
  if (primary_mask != nullptr) {  // wave-uniform: all live lanes carry primary rays of one tile
    SphereHit sh{kNoHitKey, -1};
    if (MODE == 2) sh = scan_spheres_listed_wide(sv.sph, s.o, s.d, alive, primary_mask, primary_mask2, fp.mask_words, scratch);
    if (alive) {
      if (MODE != 2) sh = scan_spheres_listed(sv.sph, s.o, s.d, primary_mask, primary_mask2, fp.mask_words);
      if (!LEAN && fp.cull_check) {  // self-check frame: the full scan must agree
        const SphereHit full = scan_spheres(sv.sph, 0u, sv.n_sph_pad - kSphBatch, s.o, s.d);
        if (full.key != sh.key || full.slot != sh.slot) atomicAdd(&fp.counters[3], 1ull);
      }
      RT_REQUEST_X2
      alive = path_shade<DEFER_SKY, LEAN>(sv, fp, s, finish_hit<MODE == 2>(sv, sh, s.o, s.d), x1, x2, seed1);
    }
    RT_PH_MARK(0)
  } else if (GRID) {
    // large scene: every lane walks the sphere grid for its own ray; rays that start outside the box the grid's
    // registration radii were derived for, or whose direction is not a unit vector (rt_grid.h), scan linearly
    SphereHit sh{kNoHitKey, -1};
    const bool near = alive & (s.o.x >= gp->near_min[0]) & (s.o.x <= gp->near_max[0]) & (s.o.y >= gp->near_min[1]) &
                      (s.o.y <= gp->near_max[1]) & (s.o.z >= gp->near_min[2]) & (s.o.z <= gp->near_max[2]) &
                      (fabsf(dot3(s.d, s.d) - 1.0f) <= 4e-6f);  // unit direction (not after a non-unit plane normal)
    if (near) sh = scan_grid(sv, *gp, s.o, s.d);
    if (alive & !near) sh = scan_spheres(sv.sph, 0u, sv.n_sph_pad - kSphBatch, s.o, s.d);
    if (!LEAN && fp.cull_check && near) {  // self-check frame: the linear scan must agree
      const SphereHit full = scan_spheres(sv.sph, 0u, sv.n_sph_pad - kSphBatch, s.o, s.d);
      if (full.key != sh.key || full.slot != sh.slot) atomicAdd(&fp.counters[3], 1ull);
    }
    RT_PH_MARK(1)
    RT_REQUEST_X2
    if (alive) alive = path_shade<DEFER_SKY, LEAN>(sv, fp, s, finish_hit(sv, sh, s.o, s.d), x1, x2, seed1);
    RT_PH_MARK(3)
  } else if (MODE == 2) {
    // secondary rays of a large scene: candidate lists, in rounds (rt_dirmask.h)
    Hit miss;
    miss.t = 0.0f;
    miss.obj = -1;
    miss.slot = -1;
    miss.plane = false;
    const Hit h = intersect_listed<LEAN>(sv, fp, s.o, s.d, alive, miss, scratch RT_PH_ARGS);
    RT_PH_MARK(1)
    RT_REQUEST_X2
    if (alive) alive = path_shade<DEFER_SKY, LEAN>(sv, fp, s, h, x1, x2, seed1);
    RT_PH_MARK(3)
  } else if (uint32_t entry = 0u; sv.dm->enabled != 0u && sv.dm->lists == 0u && dirmask_lookup(*sv.dm, s.o, s.d, alive, entry)) {
    // secondary rays of a small scene: every lane tests its own candidate set (rt_dirmask.h)
    Hit miss;
    miss.t = 0.0f;
    miss.obj = -1;
    miss.slot = -1;
    miss.plane = false;
    const Hit h = intersect_masked<LEAN>(sv, fp, s.o, s.d, alive, entry, miss);
    RT_PH_MARK(1)
    RT_REQUEST_X2
    if (alive) alive = path_shade<DEFER_SKY, LEAN>(sv, fp, s, h, x1, x2, seed1);
    RT_PH_MARK(3)
  } else if (n_live <= 32u) {
    const SphereHit sh = scan_spheres_coop(sv.sph, sv.n_sph_pad, s.o, s.d, alive, live, n_live);
    RT_PH_MARK(2)
    RT_REQUEST_X2
    if (alive) alive = path_shade<DEFER_SKY, LEAN>(sv, fp, s, finish_hit(sv, sh, s.o, s.d), x1, x2, seed1);
    RT_PH_MARK(3)
  } else {
    if (alive) {
      // the second texel is requested half-way through the scan: the first has arrived by then
#if RT_X2_PREFETCH
      const uint32_t n = sv.n_sph_pad - kSphBatch, half = (n >> 1) & ~7u;
      SphereHit sh = scan_spheres(sv.sph, 0u, half, s.o, s.d);
      RT_REQUEST_X2
      sh = scan_spheres(sv.sph, half, n, s.o, s.d, sh);
#else
      const SphereHit sh = scan_spheres(sv.sph, 0u, sv.n_sph_pad - kSphBatch, s.o, s.d);
#endif
      RT_PH_MARK(1)
      alive = path_shade<DEFER_SKY, LEAN>(sv, fp, s, finish_hit(sv, sh, s.o, s.d), x1, x2, seed1);
    }
    RT_PH_MARK(3)
  }
#undef RT_REQUEST_X2
  return alive;
}

// The ray_intersect half of one iteration for a whole wave (every lane calls it): the sphere scan the wave's state asks
// for — the tile's culled list for primary rays, the cooperative scan at <= 32 live rays, the full scan otherwise — and
// the planes.  `alive` lanes get their Hit (obj -1 = miss); the others get a miss.  Used by render_blocks, which packs the
// paths that HIT before it shades them.
template <bool LEAN, bool LISTS = false, bool TRIPS2 = (RT_LIST_TRIPS2 != 0)>
__device__ __forceinline__ Hit path_scan_wave(const SceneView &sv, const FrameParams &fp, const Path &s, bool alive,
                                              const unsigned long long *__restrict__ primary_mask,
                                              const unsigned long long *__restrict__ primary_mask2, unsigned char *scratch RT_PH_PARAMS) {
  Hit h;
  h.t = 0.0f;
  h.obj = -1;
  h.slot = -1;
  h.plane = false;
  const unsigned long long live = __builtin_amdgcn_ballot_w64(alive);
  const uint32_t n_live = static_cast<uint32_t>(__popcll(live));
  if (n_live == 0u) return h;
  if (primary_mask != nullptr) {  // wave-uniform: all live lanes carry primary rays of one tile
    SphereHit sh{kNoHitKey, -1};
    if (LISTS) sh = scan_spheres_listed_wide(sv.sph, s.o, s.d, alive, primary_mask, primary_mask2, fp.mask_words, scratch);
    if (alive) {
      if (!LISTS) sh = scan_spheres_listed(sv.sph, s.o, s.d, primary_mask, primary_mask2, fp.mask_words);
      if (!LEAN && fp.cull_check) {  // self-check frame: the full scan must agree
        const SphereHit full = scan_spheres(sv.sph, 0u, sv.n_sph_pad - kSphBatch, s.o, s.d);
        if (full.key != sh.key || full.slot != sh.slot) atomicAdd(&fp.counters[3], 1ull);
      }
      h = finish_hit<LISTS>(sv, sh, s.o, s.d);
    }
    RT_PH_MARK(0)
  } else if (LISTS) {
    // secondary rays of a large scene: candidate lists, in rounds (rt_dirmask.h)
    h = intersect_listed<LEAN, TRIPS2>(sv, fp, s.o, s.d, alive, h, scratch RT_PH_ARGS);
    RT_PH_MARK(1)
  } else if (uint32_t entry = 0u; sv.dm->enabled != 0u && sv.dm->lists == 0u && dirmask_lookup(*sv.dm, s.o, s.d, alive, entry)) {
    // secondary rays of a small scene: every lane tests its own candidate set (rt_dirmask.h)
    h = intersect_masked<LEAN>(sv, fp, s.o, s.d, alive, entry, h, scratch);
    RT_PH_MARK(1)
  } else if (n_live <= 32u) {
    const SphereHit sh = scan_spheres_coop(sv.sph, sv.n_sph_pad, s.o, s.d, alive, live, n_live);
    RT_PH_MARK(2)
    if (alive) h = finish_hit(sv, sh, s.o, s.d);
    RT_PH_MARK(3)
  } else {
    if (alive) {
      const SphereHit sh = scan_spheres(sv.sph, 0u, sv.n_sph_pad - kSphBatch, s.o, s.d);
      RT_PH_MARK(1)
      h = finish_hit(sv, sh, s.o, s.d);
    }
    RT_PH_MARK(3)
  }
  return h;
}

// frag.hlsl:60-65 + the Rgba8UnormSrgb store (D13).  At progressive_count 0 the history weight
// is exactly 0, so the history is not read.
template <bool DEFER_SKY = false, bool LEAN = false>
__device__ __forceinline__ void path_end(const FrameParams &fp, const Path &s) {
  const size_t p = s.p;
  f3 result = s.result;
  if (DEFER_SKY && (s.rough & 0x80000000u)) result = fma3(s.energy, sample_hdri(fp, s.d), result);  // frag.hlsl:14-17, D1
  if (LEAN) {  // opacity 1, no history: fma(result, 1, 0 * 0) is result itself; the sample goes to the staging buffer
    fp.rgbf32[3 * p] = result.x;
    fp.rgbf32[3 * p + 1] = result.y;
    fp.rgbf32[3 * p + 2] = result.z;
    return;
  }
  f3 prev = mk3(0.0f, 0.0f, 0.0f);
  if (fp.progressive_count != 0u) {
    if (fp.accum_mode == 1) {
      prev = mk3(fp.hist_f32[3 * p], fp.hist_f32[3 * p + 1], fp.hist_f32[3 * p + 2]);
    } else {
      const uint32_t w = fp.hist_rgba8[p];
      prev = mk3(fp.srgb_lut[w & 255u], fp.srgb_lut[(w >> 8) & 255u], fp.srgb_lut[(w >> 16) & 255u]);
    }
  }
  const f3 mixed = fma3(result, fp.opacity, prev * fp.keep);
  if (fp.rgba8)  // null in the sample pass of a batch (resolve_batch_kernel encodes)
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
  if (fp.segments) fp.segments[p] = s.seg;
  if (fp.hits)  // iterations that never ran are marked -2
    for (uint32_t i = s.seg; i < fp.reflection_limit; ++i) fp.hits[p * fp.reflection_limit + i] = -2;
}

// Number of the workgroup in launch order (the tile kernels' grids are 3-D: grid_tile_pos below).
__device__ __forceinline__ uint32_t linear_block() { return (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x; }

// Frame totals.  A workgroup reduces its four waves through LDS and issues ONE set of atomics,
// on the totals shard of its own 128-byte line (blockIdx % kTotalShards): device-scope atomics on a
// single line saturate near 90/us, and one set per wave on one line used to cost more than the
// rendering itself.  The host sums the shards (rt_get_stats / rt_get_frame_history).
__device__ __forceinline__ void add_totals(const FrameParams &fp, uint32_t seg, uint32_t rough, uint32_t iters) {
  __shared__ unsigned int block_totals[3];
  if (threadIdx.x < 3) block_totals[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    seg += __shfl_xor(seg, off, 64);
    rough += __shfl_xor(rough, off, 64);
  }
  if ((threadIdx.x & 63u) == 0) {
    atomicAdd(&block_totals[0], seg);
    atomicAdd(&block_totals[1], rough);
    atomicAdd(&block_totals[2], iters);
  }
  __syncthreads();
  if (threadIdx.x < 3 && block_totals[threadIdx.x] != 0u) {
    unsigned long long *shard = fp.counters + static_cast<size_t>(linear_block() % kTotalShards) * kTotalShardStride;
    atomicAdd(&shard[threadIdx.x], static_cast<unsigned long long>(block_totals[threadIdx.x]));
  }
}

// Tile index of the launch -> tile column / tile row inside its frame and the frame of the batch.
struct TilePos {
  uint32_t col, row, frame;
};
#ifdef RT_AB_SHAPES  // (only the resident-wave shape numbers its tiles linearly)
__device__ __forceinline__ TilePos tile_pos(const FrameParams &fp, uint32_t tile, uint32_t tiles_x) {
  TilePos t;
  const uint32_t stacked_row = tile / tiles_x;
  t.col = tile - stacked_row * tiles_x;
  t.frame = stacked_row / fp.frame_tile_rows;
  t.row = stacked_row - t.frame * fp.frame_tile_rows;
  return t;
}
#endif

// The tile kernels are launched on a 3-D grid — x: tile column / waves per workgroup, y: tile row, z: frame of the batch —
// so that a wave's tile needs no integer division (tile_pos costs ~50 vector instructions per wave: two 32-bit
// divisions of a value the compiler cannot prove uniform).
template <int WAVES_PER_BLOCK>
__device__ __forceinline__ TilePos grid_tile_pos() {
  TilePos t;
  t.col = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  t.row = blockIdx.y;
  t.frame = blockIdx.z;
  return t;
}

// The same for render_blocks, XCD-aware in launches of several frames (`banded`).  Workgroups go to the eight XCDs in turn (linear
// workgroup id % 8), so with the plain map an XCD renders every eighth 16-pixel column of the whole frame and its L2 — 4 MB — sees the
// whole candidate table (32 MB at the headline scene, 456 MB at 1 024 spheres).  Here XCD k's run of workgroups walks ONE vertical band of
// the frame (an eighth of its width) row by row: what an XCD has in flight at a time is a compact 240 x 240 pixel square of a 1080p image,
// whose secondary rays start in a few cells of the table.  The bands rotate over the XCDs every 2^RT_XCD_BAND_ROWS_LOG2 tile rows and
// every frame, so that each XCD gets its share of sky and of spheres.  A/B on one device (profiles/r04_ab_xcd_bands.txt), 8 frames per
// launch: headline +1.3 %, 3840x2160 +2.4 %, 1 024 spheres +2.8 %; rotation every 2 / 4 / 8 / 16 / 32 / 64 rows 57.3 / 57.5 / 57.6 /
// 57.8 / 57.9 / 57.7 Grays/s.  Single frames LOSE (-1 % at 2 rows ... -9 % at 32, -20 % at 64: one frame gives an XCD four and a half squares,
// not enough to even out) and keep the plain map; so does a launch whose XCDs take a tile position in all its frames before they move on
// (frames innermost, no rotation over the frames: -0.5 % ... -11 % against this map by rotation period).  Needs a grid width that is a multiple of 8 (1920 and 3840 pixels are); others: plain.
#ifndef RT_XCD_BANDS
#define RT_XCD_BANDS 1
#endif
#ifndef RT_XCD_BAND_ROWS_LOG2
#define RT_XCD_BAND_ROWS_LOG2 5
#endif
template <int WAVES_PER_BLOCK>
__device__ __forceinline__ TilePos block_tile_pos(uint32_t xcd_bands) {
  uint32_t bx = blockIdx.x, by = blockIdx.y;
  if (RT_XCD_BANDS && xcd_bands != 0u && (gridDim.x & 7u) == 0u) {
    const uint32_t l = blockIdx.y * gridDim.x + blockIdx.x;  // (a frame's workgroups are a multiple of 8: the next frame starts on XCD 0 again)
    const uint32_t k = l & 7u, i = l >> 3;
    const uint32_t bw = gridDim.x >> 3;
    by = i / bw;
    const uint32_t band = (k + (by >> RT_XCD_BAND_ROWS_LOG2) + blockIdx.z + xcd_bands - 1u) & 7u;
    bx = band * bw + (i - by * bw);
  }
  TilePos t;
  t.col = bx * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  t.row = by;
  t.frame = blockIdx.z;
  return t;
}

// STAGE: 2 = the whole scene image staged into LDS, 1 = only the sphere records (scenes of hundreds of
// spheres), 3 = the sphere records and the sphere grid (secondary rays walk the grid), 0 = nothing (scenes
// beyond a workgroup's LDS): everything else is read from global memory; 4 = nothing staged, secondary rays use the candidate
// lists of large scenes (LDS holds the waves' scan scratch).
// THREADS: 256, or 128 where the staged data is small enough for 128-thread workgroups to fill a CU's wave slots
// (a workgroup holds its LDS until its slowest wave ends: smaller groups drain faster — +10 % at 256 spheres —
// but at 1 024 spheres 28 KB per group would leave a CU with 10 waves instead of 20).
template <int STAGE, int THREADS = kTileBlock>
__global__ void __launch_bounds__(THREADS) render_tiles(DeviceScene sc, FrameParams fp) {
  extern __shared__ float4 lds_image[];
  if (STAGE != 0 && STAGE != 4) {
    const uint32_t n = STAGE == 2 ? sc.image_records : sc.n_sph_pad;
    for (uint32_t i = threadIdx.x; i < n; i += THREADS) lds_image[i] = sc.image[i];
    if (STAGE == 3) {  // the grid's cell table and item list, straight behind the sphere records
      const uint32_t m = (sc.grid.off_items - sc.grid.off_cells) + (sc.grid.n_items * 2u + 15u) / 16u;  // records
      for (uint32_t i = threadIdx.x; i < m; i += THREADS) lds_image[sc.n_sph_pad + i] = sc.image[sc.grid.off_cells + i];
    }
    __syncthreads();
  }
  SceneView sv = make_view(sc, (STAGE != 0 && STAGE != 4) ? lds_image : sc.image, STAGE == 2 ? lds_image : sc.image);
  unsigned char *scratch = STAGE == 4 ? reinterpret_cast<unsigned char *>(lds_image) : nullptr;
  constexpr int MODE = STAGE == 3 ? 1 : (STAGE == 4 ? 2 : 0);
  if (STAGE == 3) {
    sv.grid_cells = reinterpret_cast<const uint32_t *>(lds_image + sc.n_sph_pad);
    sv.grid_items = reinterpret_cast<const uint16_t *>(lds_image + sc.n_sph_pad + (sc.grid.off_items - sc.grid.off_cells));
  }
  RT_PH_DECL
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t tiles_x = (fp.width + kTile - 1) / kTile;
  const TilePos tp = grid_tile_pos<THREADS / 64>();
  const uint32_t x = tp.col * kTile + (lane & 7u);
  const uint32_t lrow = tp.row * kTile + (lane >> 3);
  uint32_t seg = 0, rough = 0;
  Path s;
  s.p = 0;
  s.seg = 0;
  s.rough = 0;
  s.o = s.d = s.energy = s.result = mk3(0.0f, 0.0f, 0.0f);
  s.rng = Rng{0.0f, 0.0f, 0.0f};
  const bool valid = (x < fp.width) & (lrow < fp.local_rows) & (tp.frame < fp.batch);
  bool alive = false;
  if (valid) {
    path_begin(fp, x, lrow, tp.frame, s);
    alive = fp.reflection_limit != 0u;
  }
  if (fp.tile_masks != nullptr && __builtin_amdgcn_ballot_w64(alive) != 0ull)
    alive = path_step_wave<true, MODE>(sv, fp, s, alive, fp.tile_masks + static_cast<size_t>(tp.col) * fp.mask_words,
                                       fp.tile_masks + static_cast<size_t>(tiles_x + tp.row) * fp.mask_words RT_PH_ARGS, &sc.grid, scratch);
  while (__builtin_amdgcn_ballot_w64(alive) != 0ull)
    alive = path_step_wave<true, MODE>(sv, fp, s, alive, nullptr, nullptr RT_PH_ARGS, &sc.grid, scratch);
  if (valid) {
    path_end<true>(fp, s);
    seg = s.seg;
    rough = s.rough & 0x7fffffffu;
  }
  add_totals(fp, seg, rough, 0u);
}

// Block-compacting tile kernel.  A workgroup of NW waves starts NW 8x8 tiles (one pixel per lane).
// Every bounce, after the scan, the paths of the whole workgroup that HIT something are counted (ballot +
// popcount per wave, summed through LDS); as soon as they fit into fewer waves they are packed through an
// LDS exchange buffer (80 B each: path state + the hit) into waves 0..k-1 and the surplus waves END — gfx950's
// s_barrier drops ended waves from its count (tools/microbench/barrier_exit.hip), and their registers and
// wave slots go to the next workgroup — then the hits are shaded.  Path lengths vary 1..R, so without this
// most of a tile's later scans run with a handful of live lanes.  Pixel results do not depend on which
// lane finishes a path, so the frame stays bit-identical.
// LISTS: a large scene — nothing is staged (records and materials come from memory through the vector cache: a thousand spheres'
// records alone would cost the workgroups their occupancy), secondary rays use the candidate lists; LDS holds the exchange buffer
// and the two waves' list-scan scratch.
template <int NW, int WAVES, bool LEAN, bool SPILL = false, bool LISTS = false>
__global__ void __launch_bounds__(NW * 64, WAVES) render_blocks(DeviceScene sc, FrameParams fp) {
  extern __shared__ float4 lds_all[];
  // The single-frame form (WAVES = 5) overlaps with its neighbours' launches and runs best at 5 waves per SIMD (A/B on
  // one device: 34.3 against 31.8 - 33.3 Grays/s at 6; holding the occupancy with a larger LDS request instead: 30.1).
  // Its code fits the 80 registers of 6, so the register count is lifted to the next occupancy step by naming v88 as
  // clobbered — no instruction, the allocation just becomes 96.
  if (WAVES == 5) asm volatile("; 5 waves per SIMD: register count held above 88" ::: "v88");
  // Live-path counts of the waves, double-buffered by iteration parity: iteration k writes and reads
  // wave_count[k & 1].  A wave reaches its write of iteration k+2 (same buffer) only after the count
  // barrier of iteration k+1, which every live wave joins after it has finished its iteration-k reads
  // (program order), so no write can pass a read of the value it replaces — also on iterations that
  // do not pack and therefore have no second barrier.
  __shared__ unsigned int wave_count[2][NW];
  float4 *xchg = lds_all + (LISTS ? 0u : sc.image_records);  // kXchgPaths paths x 4 float4, behind the staged scene image
  // the waves' scratch for the balanced direction-mask scan (shares the exchange buffer's LDS when NW == 2: kScratchInXchg);
  // wave w's part starts w * kScanScratchBytes (kListScratchBytes) in
  unsigned char *scratch = reinterpret_cast<unsigned char *>(xchg) + (scratch_in_xchg(NW) ? 0u : xchg_bytes(NW));
  if (!LISTS)
    for (uint32_t i = threadIdx.x; i < sc.image_records; i += NW * 64) lds_all[i] = sc.image[i];
  const SceneView sv = LISTS ? make_view(sc, sc.image, sc.image) : make_view(sc, lds_all, lds_all);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  const uint32_t tiles_x = (fp.width + kTile - 1) / kTile;
  const TilePos tp = block_tile_pos<NW>(fp.xcd_bands);
  const uint32_t x = tp.col * kTile + (lane & 7u);
  const uint32_t lrow = tp.row * kTile + (lane >> 3);
  Path s;
  s.p = 0;
  s.seg = 0;
  s.rough = 0;
  s.o = s.d = s.energy = s.result = mk3(0.0f, 0.0f, 0.0f);
  s.rng = Rng{0.0f, 0.0f, 0.0f};
  bool alive = false;
  uint32_t seg_total = 0, rough_total = 0, iters = 0;
  if ((x < fp.width) & (lrow < fp.local_rows) & (tp.frame < fp.batch)) {
    path_begin(fp, x, lrow, tp.frame, s);
    alive = fp.reflection_limit != 0u;
    if (!alive) path_end<RT_DEFER_SKY_BLOCKS, LEAN>(fp, s);
  }
  uint32_t live_waves = NW;
  RT_PH_DECL
  __syncthreads();  // sphere records staged
  RT_PH_MARK(7)
  const unsigned long long *primary_mask =
      fp.tile_masks != nullptr ? fp.tile_masks + static_cast<size_t>(tp.col) * fp.mask_words : nullptr;
  const unsigned long long *primary_mask2 =
      fp.tile_masks != nullptr ? fp.tile_masks + static_cast<size_t>(tiles_x + tp.row) * fp.mask_words : nullptr;
  // A finished path keeps its result in the lane's registers and is written out (sRGB encode +
  // stores, ~150 instructions) together with the other finished lanes when the lane is about to be
  // re-used or the wave ends, not one partial-wave call per bounce.
  bool finished = false;
  // Every bounce: scan -> the paths that missed end -> the workgroup counts the paths that HIT and packs them into fewer
  // waves when they fit -> shade.  Packing between the scan and the shading (instead of after the shading) spares the
  // surplus wave its shading pass: the misses, a third of the rays of a bounce, have left by then.
  for (;;) {
    // Issue priority by phase: everything but the shading runs at priority 1.  The scan, the count and the re-packing are chains
    // of dependent LDS and memory operations with little arithmetic between them, the shading is arithmetic with plenty of
    // independent work: when the scheduler prefers the waves in their chains, those get through their round trips sooner and
    // the shading waves fill the issue slots in between (A/B on one device, here and in tail_stream: +1.4 %; the scan alone
    // raised: +0.9 %; the shading raised instead: -0.3 %; three levels, scan above the rest: +0.9 % — profiles/r03_tail_spill.txt)
    __builtin_amdgcn_s_setprio(1);
    // Large scenes (nothing staged: the kernel waits on memory, not on its issue slots): the two random() texels of the shading are
    // requested around the scan — the first before it, the second (its address needs the first one's value) right after it —
    // instead of one after the other at the head of the shading.  The RNG state does not change in between.
    constexpr bool EARLY_X = LISTS && RT_LIST_EARLY_X;
    float x1 = 0.0f, x2 = 0.0f, seed1 = 0.0f;
    if (EARLY_X && alive) x1 = fp.rnd[rng_texel(fp, s.rng)];
    Hit h = path_scan_wave<LEAN, LISTS>(sv, fp, s, alive, primary_mask, primary_mask2, scratch RT_PH_ARGS);
    primary_mask = nullptr;  // only the first segment is a primary ray of this wave's tile
    if (alive && h.obj < 0) {  // frag.hlsl:14-17: a miss ends the path (its skybox texel is fetched in path_end)
      (void)path_shade<RT_DEFER_SKY_BLOCKS, LEAN>(sv, fp, s, h, 0.0f, 0.0f, 0.0f);
      alive = false;
      finished = true;
    }
    if (EARLY_X && alive) {
      seed1 = rng_next_seed(s.rng.seed, x1);
      x2 = fp.rnd[rng_texel(fp, Rng{seed1, s.rng.px, s.rng.py})];
    }
    unsigned int *counts = wave_count[iters & 1u];
    iters += 1u;
    const unsigned long long live = __builtin_amdgcn_ballot_w64(alive);
    if (lane == 0) counts[wave] = static_cast<uint32_t>(__popcll(live));
    __syncthreads();
    RT_PH_MARK(4)
    uint32_t total = 0, before = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const uint32_t c = (static_cast<uint32_t>(w) < live_waves) ? counts[w] : 0u;
      before += (static_cast<uint32_t>(w) < wave) ? c : 0u;
      total += c;
    }
    if (total == 0u) break;
    if (SPILL && total <= fp.spill_max && iters < fp.reflection_limit) {
      // Tail spill: so few paths are left that the rest of their bounces would run in a mostly empty wave.  They go — path
      // state + hit, the record of the exchange buffer — to the launch's path queue, where tail_stream picks them up into
      // full waves, and the workgroup ends.  (When only the last shading is left, iters == reflection_limit, it is done here.)
      if (live != 0ull) {  // wave-uniform
        const uint32_t shard = linear_block() % kQueueShards;
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&fp.tile_counters[shard * 32u], static_cast<uint32_t>(__popcll(live)));
        base = __builtin_amdgcn_readfirstlane(base);
        if (alive) {
          const size_t r = static_cast<size_t>(shard) * fp.queue_shard_cap + base +
                           __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(live >> 32),
                                                     __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(live), 0u));
          float4 *q = fp.queue + r;
          q[0] = make_float4(s.o.x, s.o.y, s.o.z, s.d.x);
          q[fp.queue_plane] = make_float4(s.d.y, s.d.z, s.energy.x, s.energy.y);
          q[2 * fp.queue_plane] = make_float4(s.energy.z, s.result.x, s.result.y, s.result.z);
          q[3 * fp.queue_plane] = make_float4(s.rng.seed, s.rng.px, s.rng.py, __uint_as_float(s.p));
          q[4 * fp.queue_plane] = make_float4(h.t, __int_as_float(h.slot), __int_as_float(h.obj),
                                              __uint_as_float((h.plane ? 1u : 0u) | (s.seg << 1)));
          alive = false;  // not finished: the path goes on elsewhere
        }
      }
      break;
    }
    const uint32_t need = (total + 63u) >> 6;
    if (need < live_waves && total <= xchg_paths(NW)) {
      // pack: path number = paths in lower waves + live lanes below this one; the hit travels with the path
      if (alive) {
        const uint32_t r = before + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(live >> 32),
                                                              __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(live), 0u));
        float4 *q = xchg + 5u * r;
        q[0] = make_float4(s.o.x, s.o.y, s.o.z, s.d.x);
        q[1] = make_float4(s.d.y, s.d.z, s.energy.x, s.energy.y);
        q[2] = make_float4(s.energy.z, s.result.x, s.result.y, s.result.z);
        q[3] = make_float4(s.rng.seed, s.rng.px, s.rng.py, __uint_as_float(s.p));
        q[4] = make_float4(h.t, __int_as_float(h.slot), __int_as_float(h.obj), __uint_as_float((h.plane ? 1u : 0u) | (s.seg << 1)));
      }
      if (finished) {  // the lane's registers are about to change hands
        path_end<RT_DEFER_SKY_BLOCKS, LEAN>(fp, s);
        seg_total += s.seg;
        finished = false;
      }
      __syncthreads();
      if (wave >= need) break;  // surplus wave: its paths now live in a lower wave
      const uint32_t r = wave * 64u + lane;
      alive = r < total;
      if (alive) {
        const float4 *q = xchg + 5u * r;
        const float4 a = q[0], b = q[1], c = q[2], e = q[3], g = q[4];
        s.o = mk3(a.x, a.y, a.z);
        s.d = mk3(a.w, b.x, b.y);
        s.energy = mk3(b.z, b.w, c.x);
        s.result = mk3(c.y, c.z, c.w);
        s.rng = Rng{e.x, e.y, e.z};
        s.p = __float_as_uint(e.w);  // the whole word: a launch may carry up to 2^32 pixels (64 frames of 1080p are 2^27)
        s.seg = __float_as_uint(g.w) >> 1;
        s.rough = 0;
        h.t = g.x;
        h.slot = __float_as_int(g.y);
        h.obj = __float_as_int(g.z);
        h.plane = (__float_as_uint(g.w) & 1u) != 0u;
        if (EARLY_X) {  // the lane carries another path now: its texels
          x1 = fp.rnd[rng_texel(fp, s.rng)];
          seed1 = rng_next_seed(s.rng.seed, x1);
          x2 = fp.rnd[rng_texel(fp, Rng{seed1, s.rng.px, s.rng.py})];
        }
      }
      live_waves = need;
      // No barrier here: the exchange buffer is next written after the next count barrier, which the
      // surviving waves only pass once their reads above have completed.
      RT_PH_MARK(5)
    }
    __builtin_amdgcn_s_setprio(0);
    if (alive) {  // shade the hit; the first random() texel is requested here, by the lanes that still carry a path
      if (!EARLY_X) x1 = fp.rnd[rng_texel(fp, s.rng)];
      alive = path_shade<RT_DEFER_SKY_BLOCKS, LEAN, EARLY_X>(sv, fp, s, h, x1, x2, seed1);
      rough_total += s.rough & 0x7fffffffu;
      s.rough &= 0x80000000u;
      finished |= !alive;
    }
    RT_PH_MARK(3)
  }
  RT_PH_MARK(5)
  if (finished) {
    path_end<RT_DEFER_SKY_BLOCKS, LEAN>(fp, s);
    seg_total += s.seg;
  }
  RT_PH_MARK(6)
  // totals: one atomic set per wave on the workgroup's shard (waves end at different times here)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    seg_total += __shfl_xor(seg_total, off, 64);
    rough_total += __shfl_xor(rough_total, off, 64);
  }
  if (lane == 0) {
    unsigned long long *shard = fp.counters + static_cast<size_t>(linear_block() % kTotalShards) * kTotalShardStride;
    if (seg_total) atomicAdd(&shard[0], static_cast<unsigned long long>(seg_total));
    if (rough_total) atomicAdd(&shard[1], static_cast<unsigned long long>(rough_total));
    atomicAdd(&shard[2], static_cast<unsigned long long>(iters));
  }
#ifndef RT_PH_TAIL_ONLY  // (diagnostic build of tail_stream's stamps alone)
  RT_PH_FLUSH(fp.counters + static_cast<size_t>(linear_block() % kTotalShards) * kTotalShardStride)
#endif
}

// ---- launch shapes of finished A/Bs (RT_OPT_VARIANT 2 and 4): compiled only with -DRT_AB_SHAPES (`make ab` -> librt355_ab.so, which the
// A/B tools and tests/test_ab_shapes_gpu.py use); the shipped library maps both to the re-packing shape -------------------------------
#ifdef RT_AB_SHAPES
// Streamed shape (variant 4), for launches of several frames that write float samples only.  Two kernels:
//   primary_stream  one 8x8 tile per wave: the primary segment with the tile's culled sphere list, exactly as
//                   render_blocks' first iteration; paths that end there are written out, the others are appended
//                   to a path queue in global memory (64 B of state each, four float4 planes; one atomic per wave on
//                   the write head of the workgroup's shard);
//   bounce_stream   resident waves take chunks of paths off the queue and REFILL lanes whose path has ended from
//                   it after every bounce, so scans run with full waves until the queue is dry — no per-bounce
//                   barrier, no workgroup re-packing; waves synchronise once (scene image staged into LDS).
// Pixel results do not depend on which lane carries a path, so the frame stays bit-identical.  Needs many paths per
// resident wave (multi-frame launches; a single 1080p frame has ~200).  Its refill waits on four 16-byte loads per
// bounce (16 % of a wave's time) — but prefetching the records into an LDS ring with global_load_lds_dwordx4
// (tools/microbench/lds_dma.hip) made it 1 % SLOWER: the SIMDs are issue-bound, another wave always has work, and
// the ring's bookkeeping is more instructions.
#ifndef RT_STREAM_NW1
#define RT_STREAM_NW1 4  // waves (tiles) per workgroup of the primary pass
#endif
#ifndef RT_STREAM_NW2
#define RT_STREAM_NW2 2  // waves per workgroup of the bounce pass (they share the staged scene image only)
#endif
#ifndef RT_STREAM_MIN_TILES
#define RT_STREAM_MIN_TILES 0  // auto rule: sample-only launches of this many 8x8 tiles are streamed; 0 = never (A/B on one device,
                               // 1080p, frames per launch 2 / 4 / 8 / 16: streamed 26.4 / 31.6 / 33.6 / 34.2 Grays/s against 30.4 /
                               // 32.8 / 34.1 / 34.0 re-packed — fuller waves take the root for 18 % of the spheres they test
                               // instead of 10 %, shade with every branch taken, and the queue costs instructions of its own)
#endif
#ifndef RT_STREAM_WAVES
#define RT_STREAM_WAVES 6  // waves per SIMD the bounce pass is budgeted for
#endif
template <int NW>
__global__ void __launch_bounds__(NW * 64) primary_stream(DeviceScene sc, FrameParams fp) {
  extern __shared__ float4 lds_all[];
  for (uint32_t i = threadIdx.x; i < sc.image_records; i += NW * 64) lds_all[i] = sc.image[i];
  const SceneView sv = make_view(sc, lds_all, lds_all);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t tiles_x = (fp.width + kTile - 1) / kTile;
  const TilePos tp = grid_tile_pos<NW>();
  const uint32_t x = tp.col * kTile + (lane & 7u);
  const uint32_t lrow = tp.row * kTile + (lane >> 3);
  Path s;
  s.p = 0;
  s.seg = 0;
  s.rough = 0;
  s.o = s.d = s.energy = s.result = mk3(0.0f, 0.0f, 0.0f);
  s.rng = Rng{0.0f, 0.0f, 0.0f};
  bool alive = false;
  uint32_t seg_total = 0;
  if ((x < fp.width) & (lrow < fp.local_rows) & (tp.frame < fp.batch)) {
    path_begin(fp, x, lrow, tp.frame, s);
    alive = fp.reflection_limit != 0u;
    if (!alive) path_end<false, true>(fp, s);
  }
  RT_PH_DECL
  __syncthreads();  // scene image staged
  const unsigned long long *primary_mask =
      fp.tile_masks != nullptr ? fp.tile_masks + static_cast<size_t>(tp.col) * fp.mask_words : nullptr;
  const unsigned long long *primary_mask2 =
      fp.tile_masks != nullptr ? fp.tile_masks + static_cast<size_t>(tiles_x + tp.row) * fp.mask_words : nullptr;
  const bool was = alive;
  alive = path_step_wave<false, false, true>(sv, fp, s, alive, primary_mask, primary_mask2 RT_PH_ARGS);
  if (was & !alive) {
    path_end<false, true>(fp, s);
    seg_total = s.seg;
  }
  const unsigned long long live = __builtin_amdgcn_ballot_w64(alive);
  if (live != 0ull) {  // wave-uniform
    const uint32_t shard = linear_block() % kQueueShards;
    uint32_t base = 0;
    if (lane == 0) base = atomicAdd(&fp.tile_counters[shard * 32u], static_cast<uint32_t>(__popcll(live)));
    base = __builtin_amdgcn_readfirstlane(base);
    if (alive) {
      const size_t r = static_cast<size_t>(shard) * fp.queue_shard_cap + base +
                       __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(live >> 32),
                                                 __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(live), 0u));
      float4 *q = fp.queue + r;
      q[0] = make_float4(s.o.x, s.o.y, s.o.z, s.d.x);
      q[fp.queue_plane] = make_float4(s.d.y, s.d.z, s.energy.x, s.energy.y);
      q[2 * fp.queue_plane] = make_float4(s.energy.z, s.result.x, s.result.y, s.result.z);
      q[3 * fp.queue_plane] = make_float4(s.rng.seed, s.rng.px, s.rng.py, __uint_as_float(s.p | (s.seg << 26)));
    }
  }
  add_totals(fp, seg_total, s.rough & 0x7fffffffu, 0u);
  RT_PH_FLUSH(fp.counters + static_cast<size_t>(linear_block() % kTotalShards) * kTotalShardStride)
}

template <int NW, int WAVES>
__global__ void __launch_bounds__(NW * 64, WAVES) bounce_stream(DeviceScene sc, FrameParams fp, uint32_t refill_min, uint32_t chunk) {
  extern __shared__ float4 lds_all[];
  for (uint32_t i = threadIdx.x; i < sc.image_records; i += NW * 64) lds_all[i] = sc.image[i];
  __syncthreads();
  const SceneView sv = make_view(sc, lds_all, lds_all);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  // Work distribution: a wave takes `chunk` consecutive records off the read head of a shard (one atomic), starting
  // with the shard of its number and moving on when a shard is dry.  The write heads are final: the primary pass
  // ran before this kernel on the same stream.
  uint32_t shard = (blockIdx.x * NW + wave) % kQueueShards, shards_done = 0;
  size_t q_pos = 0, q_end = 0;  // records of the chunk in hand [q_pos, q_end), absolute record numbers
  Path s;
  s.p = 0;
  s.seg = 0;
  s.rough = 0;
  s.o = s.d = s.energy = s.result = mk3(0.0f, 0.0f, 0.0f);
  s.rng = Rng{0.0f, 0.0f, 0.0f};
  bool alive = false;
  bool work_left = true;
  uint32_t seg_total = 0, rough_total = 0, wave_iters = 0;
  RT_PH_DECL
  for (;;) {
    const unsigned long long idle = __builtin_amdgcn_ballot_w64(!alive);
    const uint32_t n_idle = static_cast<uint32_t>(__popcll(idle));
    if (work_left && n_idle >= refill_min) {
      unsigned long long need = idle;
      while (need != 0ull) {
        if (q_pos >= q_end) {  // next chunk
          bool got = false;
          while (shards_done < kQueueShards) {
            const uint32_t filled = fp.tile_counters[shard * 32u];
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(&fp.tile_counters[(kQueueShards + shard) * 32u], chunk);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t < filled) {
              q_pos = static_cast<size_t>(shard) * fp.queue_shard_cap + t;
              q_end = q_pos + min(chunk, filled - t);
              got = true;
              break;
            }
            shard = (shard + 1u) % kQueueShards;  // this shard is dry for good
            shards_done += 1u;
          }
          if (!got) {
            work_left = false;
            break;
          }
        }
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(need >> 32),
                                                         __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(need), 0u));
        const bool needy = (need >> lane) & 1ull;
        const uint32_t n_need = static_cast<uint32_t>(__popcll(need));
        const uint32_t room = static_cast<uint32_t>(q_end - q_pos);
        const uint32_t taken = n_need < room ? n_need : room;
        if (needy && below < taken) {
          const float4 *q = fp.queue + q_pos + below;
          const float4 a = q[0], b = q[fp.queue_plane], c = q[2 * fp.queue_plane], e = q[3 * fp.queue_plane];
          s.o = mk3(a.x, a.y, a.z);
          s.d = mk3(a.w, b.x, b.y);
          s.energy = mk3(b.z, b.w, c.x);
          s.result = mk3(c.y, c.z, c.w);
          s.rng = Rng{e.x, e.y, e.z};
          const uint32_t packed = __float_as_uint(e.w);
          s.p = packed & 0x03ffffffu;
          s.seg = packed >> 26;
          s.rough = 0;
          alive = true;
        }
        q_pos += taken;
        need = __builtin_amdgcn_ballot_w64(needy && below >= taken);
      }
    }
    RT_PH_MARK(5)
    if (__builtin_amdgcn_ballot_w64(alive) == 0ull) {
      if (!work_left) break;
      continue;
    }
    wave_iters += 1u;
    const bool was = alive;
    alive = path_step_wave<false, false, true>(sv, fp, s, alive, nullptr, nullptr RT_PH_ARGS);
    rough_total += s.rough & 0x7fffffffu;
    s.rough = 0;
    if (was & !alive) {
      path_end<false, true>(fp, s);
      seg_total += s.seg;
    }
    RT_PH_MARK(6)
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    seg_total += __shfl_xor(seg_total, off, 64);
    rough_total += __shfl_xor(rough_total, off, 64);
  }
  if (lane == 0) {
    unsigned long long *shard_totals = fp.counters + static_cast<size_t>(blockIdx.x % kTotalShards) * kTotalShardStride;
    if (seg_total) atomicAdd(&shard_totals[0], static_cast<unsigned long long>(seg_total));
    if (rough_total) atomicAdd(&shard_totals[1], static_cast<unsigned long long>(rough_total));
    if (wave_iters) atomicAdd(&shard_totals[2], static_cast<unsigned long long>(wave_iters));
  }
  RT_PH_FLUSH(fp.counters + static_cast<size_t>(blockIdx.x % kTotalShards) * kTotalShardStride)
}

#endif  // RT_AB_SHAPES

// Tail of the re-packing shape (FrameParams::spill_max): render_blocks' workgroups hand the last few paths of their tiles —
// the ones that would otherwise bounce on in a mostly empty wave, each with the hit its scan has just found — to the path
// queue; resident waves here take them off it in chunks, SHADE them (every lane that carries a path carries a pending hit),
// scan the survivors' next segment, and refill the lanes whose path has ended.  The expensive half of an iteration, the
// shading, therefore runs with full waves until the queue is dry.  Pixel results do not depend on which lane carries a
// path, so the frame stays bit-identical.  Runs after render_blocks on the same stream: the write heads are final.
#ifndef RT_TAIL_NW
#define RT_TAIL_NW 2  // waves per workgroup (they share the staged scene image only)
#endif
#ifndef RT_TAIL_WAVES
#define RT_TAIL_WAVES 5  // waves per SIMD tail_stream is budgeted for (96 registers, no scratch; 6: 52 B, 7: 92 B of scratch per lane —
                         // A/B on one device at threshold 32: 53.0 / 51.8 / 49.3 Grays/s)
#endif
// tail_stream on a scene with candidate lists: what counts there is the number of waves that wait side by side (it is a third of a
// 1 024-sphere launch and issues a quarter of the time), not the work in flight inside one: ONE pair per lane and trip and the random()
// texel requested at the shading — twelve registers fewer — fit 5 waves per SIMD with 12 B of scratch.  A/B on one device, 1 024
// spheres, two launches in flight / one: 4 waves with two pairs + early texel 40.0 / 38.5 Grays/s, 5 waves with them (36 B of scratch)
// 40.3 / 37.8, 5 waves without 41.2 / 38.7, 6 waves without (84 B) 37.2 / 35.2.
#ifndef RT_TAIL_TRIPS2
#define RT_TAIL_TRIPS2 0
#endif
#ifndef RT_TAIL_EARLY_X
#define RT_TAIL_EARLY_X 0
#endif
#ifndef RT_TAIL_WAVES_LISTS
#define RT_TAIL_WAVES_LISTS 5
#endif
template <int NW, int WAVES, bool LISTS = false>
__global__ void __launch_bounds__(NW * 64, WAVES) tail_stream(DeviceScene sc, FrameParams fp, uint32_t refill_min, uint32_t chunk) {
  extern __shared__ float4 lds_all[];
  if (!LISTS) {
    for (uint32_t i = threadIdx.x; i < sc.image_records; i += NW * 64) lds_all[i] = sc.image[i];
    __syncthreads();
  }
  const SceneView sv = LISTS ? make_view(sc, sc.image, sc.image) : make_view(sc, lds_all, lds_all);
  unsigned char *scratch = reinterpret_cast<unsigned char *>(lds_all + (LISTS ? 0u : sc.image_records));  // NW x kScanScratchBytes (kListScratchBytes)
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;
  // Every wave draws on ONE shard of the queue, its own: the workgroups of render_blocks append to the shards in turn and
  // the waves here are dealt over them evenly, so the shards run dry together — and a wave learns that it is done from one
  // failed bump of one counter.  (Looking into the other shards as well — one by one, all at once, or only when the own one
  // is dry — costs each of the ~5 000 waves a visit to each of the 32 busy lines at the end, and a line serves ~90 visits
  // per microsecond: 30 - 170 microseconds per launch, measured: profiles/r03_tail_spill.txt.)
  const uint32_t shard = (blockIdx.x * NW + wave) % kQueueShards;
  const uint32_t own_filled = fp.tile_counters[shard * 32u];  // final: render_blocks ran before this kernel
  bool own_dry = own_filled == 0u;
  size_t q_pos = 0, q_end = 0;  // records of the chunk in hand [q_pos, q_end), absolute record numbers
  Path s;
  s.p = 0;
  s.seg = 0;
  s.rough = 0;
  s.o = s.d = s.energy = s.result = mk3(0.0f, 0.0f, 0.0f);
  s.rng = Rng{0.0f, 0.0f, 0.0f};
  Hit h;
  h.t = 0.0f;
  h.obj = -1;
  h.slot = -1;
  h.plane = false;
  bool alive = false;     // the lane carries a path whose hit `h` is to be shaded
  bool finished = false;  // the lane holds the result of a path that has ended (written out before the lane is re-used)
  bool work_left = true;
  // large scenes (the kernel waits on memory): a surviving path's first random() texel is requested right after its scan, a whole
  // refill ahead of the shading that uses it
  constexpr bool EARLY_X = LISTS && RT_TAIL_EARLY_X;
  float x1_early = 0.0f;
  bool has_x1 = false;
  uint32_t seg_total = 0, rough_total = 0, wave_iters = 0;
  RT_PH_DECL
  for (;;) {
    const unsigned long long idle = __builtin_amdgcn_ballot_w64(!alive);
    const uint32_t n_idle = static_cast<uint32_t>(__popcll(idle));
    if (work_left && n_idle >= refill_min) {
      if (finished) {  // the lane's registers are about to change hands
        path_end<RT_DEFER_SKY_BLOCKS, true>(fp, s);
        seg_total += s.seg;
        finished = false;
      }
      RT_PH_MARK(7)
      unsigned long long need = idle;
      while (need != 0ull) {
        if (q_pos >= q_end) {  // next chunk
          bool got = false;
          if (!own_dry) {
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(&fp.tile_counters[(kQueueShards + shard) * 32u], chunk);
            t = __builtin_amdgcn_readfirstlane(t);
            if (t < own_filled) {
              q_pos = static_cast<size_t>(shard) * fp.queue_shard_cap + t;
              q_end = q_pos + min(chunk, own_filled - t);
              got = true;
            } else {
              own_dry = true;
            }
          }
          RT_PH_MARK(4)
          if (!got) {
            work_left = false;
            break;
          }
        }
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(need >> 32),
                                                         __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(need), 0u));
        const bool needy = (need >> lane) & 1ull;
        const uint32_t n_need = static_cast<uint32_t>(__popcll(need));
        const uint32_t room = static_cast<uint32_t>(q_end - q_pos);
        const uint32_t taken = n_need < room ? n_need : room;
        if (needy && below < taken) {
          const float4 *q = fp.queue + q_pos + below;
          const float4 a = q[0], b = q[fp.queue_plane], c = q[2 * fp.queue_plane], e = q[3 * fp.queue_plane],
                       g = q[4 * fp.queue_plane];
          s.o = mk3(a.x, a.y, a.z);
          s.d = mk3(a.w, b.x, b.y);
          s.energy = mk3(b.z, b.w, c.x);
          s.result = mk3(c.y, c.z, c.w);
          s.rng = Rng{e.x, e.y, e.z};
          s.p = __float_as_uint(e.w);
          s.seg = __float_as_uint(g.w) >> 1;
          s.rough = 0;
          h.t = g.x;
          h.slot = __float_as_int(g.y);
          h.obj = __float_as_int(g.z);
          h.plane = (__float_as_uint(g.w) & 1u) != 0u;
          alive = true;
          has_x1 = false;
        }
        q_pos += taken;
        need = __builtin_amdgcn_ballot_w64(needy && below >= taken);
      }
    }
    RT_PH_MARK(5)
#ifdef RT_PROFILE_PHASES  // how long the records take to arrive, apart from the shading
    __builtin_amdgcn_s_waitcnt(0);
    RT_PH_MARK(0)
#endif
    if (__builtin_amdgcn_ballot_w64(alive) == 0ull) {
      if (!work_left) break;
      continue;
    }
    wave_iters += 1u;
    __builtin_amdgcn_s_setprio(0);  // issue priority by phase, as in render_blocks: the shading below everything else
    if (alive) {  // shade the pending hit (render_blocks' shading step)
      float x1 = x1_early;
      if (!EARLY_X || !has_x1) x1 = fp.rnd[rng_texel(fp, s.rng)];
      alive = path_shade<RT_DEFER_SKY_BLOCKS, true>(sv, fp, s, h, x1, 0.0f, 0.0f);
      rough_total += s.rough & 0x7fffffffu;
      s.rough &= 0x80000000u;
      finished |= !alive;
    }
    RT_PH_MARK(3)
    __builtin_amdgcn_s_setprio(1);
    h = path_scan_wave<true, LISTS, (RT_TAIL_TRIPS2 != 0)>(sv, fp, s, alive, nullptr, nullptr, scratch RT_PH_ARGS);
    if (alive && h.obj < 0) {  // frag.hlsl:14-17: a miss ends the path (its skybox texel is fetched in path_end)
      (void)path_shade<RT_DEFER_SKY_BLOCKS, true>(sv, fp, s, h, 0.0f, 0.0f, 0.0f);
      alive = false;
      finished = true;
    }
    if (EARLY_X) {
      has_x1 = alive;
      if (alive) x1_early = fp.rnd[rng_texel(fp, s.rng)];
    }
    RT_PH_MARK(6)
  }
  if (finished) {
    path_end<RT_DEFER_SKY_BLOCKS, true>(fp, s);
    seg_total += s.seg;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    seg_total += __shfl_xor(seg_total, off, 64);
    rough_total += __shfl_xor(rough_total, off, 64);
  }
  if (lane == 0) {
    unsigned long long *shard_totals = fp.counters + static_cast<size_t>(blockIdx.x % kTotalShards) * kTotalShardStride;
    if (seg_total) atomicAdd(&shard_totals[0], static_cast<unsigned long long>(seg_total));
    if (rough_total) atomicAdd(&shard_totals[1], static_cast<unsigned long long>(rough_total));
    if (wave_iters) atomicAdd(&shard_totals[2], static_cast<unsigned long long>(wave_iters));
  }
  RT_PH_FLUSH(fp.counters + static_cast<size_t>(blockIdx.x % kTotalShards) * kTotalShardStride)
}

#ifdef RT_AB_SHAPES
// Resident waves + lane refill.  `refill_min`: idle lanes a wave tolerates before it pulls
// new pixels (1 = refill at once).  fp.tile_counters: 8 counters, 128 B apart, zero at launch.
__global__ void __launch_bounds__(kBlock, RT_PERSISTENT_WAVES) render_persistent(DeviceScene sc, FrameParams fp, uint32_t refill_min,
                                                            uint32_t tile_chunk) {
  extern __shared__ float4 lds_image[];
  for (uint32_t i = threadIdx.x; i < sc.image_records; i += kBlock) lds_image[i] = sc.image[i];
  __syncthreads();
  const SceneView sv = make_view(sc, lds_image, lds_image);
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t tiles_x = (fp.width + kTile - 1) / kTile;
  const uint32_t n_tiles = tiles_x * fp.frame_tile_rows * fp.batch;
  // Work distribution: the frame's tiles are cut into chunks of `tile_chunk` consecutive tiles;
  // chunk c belongs to shard c % 8, each shard has its own counter on its own 128-byte line, a
  // wave starts on the shard of its workgroup (workgroups are dealt round-robin over the 8 XCDs)
  // and moves on to the next shard when its own runs dry.  One device-scope counter saturates
  // near 90 dequeues/us (MI355X_MICROARCH.md, row "dequeue"), which a 1080p frame's 32,400
  // tiles would hit; sharded, chunked counters do not.
  constexpr uint32_t kShards = 8;
  const uint32_t n_chunks = (n_tiles + tile_chunk - 1u) / tile_chunk;
  uint32_t shard = blockIdx.x % kShards, shards_done = 0;
  uint32_t chunk_end = 0;  // one past the last tile of the chunk in hand

  Path s;
  s.p = 0;
  s.seg = 0;
  s.rough = 0;
  s.o = s.d = s.energy = s.result = mk3(0.0f, 0.0f, 0.0f);
  s.rng = Rng{0.0f, 0.0f, 0.0f};
  bool alive = false;
  uint32_t tile = 0, tpos = 64;  // wave-uniform cursor into the current tile
  bool work_left = n_tiles != 0u;
  uint32_t seg_total = 0, rough_total = 0, wave_iters = 0;
  RT_PH_DECL

  for (;;) {
    const unsigned long long idle = __builtin_amdgcn_ballot_w64(!alive);
    const uint32_t n_idle = static_cast<uint32_t>(__popcll(idle));
    if (work_left && (n_idle >= refill_min || n_idle == 64u)) {
      unsigned long long need = idle;
      while (need != 0ull) {
        if (tpos >= 64u) {  // next tile: from the chunk in hand, else a new chunk off the counters
          if (tile + 1u < chunk_end) {
            tile += 1u;
          } else {
            bool got = false;
            while (shards_done < kShards) {
              uint32_t t = 0;
              if (lane == 0) t = atomicAdd(&fp.tile_counters[shard * 32u], 1u);
              t = __builtin_amdgcn_readfirstlane(t);
              const uint32_t c = t * kShards + shard;
              if (c < n_chunks) {
                tile = c * tile_chunk;
                chunk_end = min(tile + tile_chunk, n_tiles);
                got = true;
                break;
              }
              shard = (shard + 1u) % kShards;  // this shard is exhausted for good
              shards_done += 1u;
            }
            if (!got) {
              work_left = false;
              break;
            }
          }
          tpos = 0;
        }
        // rank of this lane among the lanes that still need a pixel
        const uint32_t below = __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(need >> 32),
                                                         __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(need), 0u));
        const bool needy = (need >> lane) & 1ull;
        const uint32_t k = tpos + below;
        if (needy && k < 64u) {
          const TilePos tp = tile_pos(fp, tile, tiles_x);
          const uint32_t x = tp.col * kTile + (k & 7u);
          const uint32_t lrow = tp.row * kTile + (k >> 3);
          if ((x < fp.width) & (lrow < fp.local_rows)) {
            path_begin(fp, x, lrow, tp.frame, s);
            alive = fp.reflection_limit != 0u;
            if (!alive) path_end<RT_DEFER_SKY_BLOCKS>(fp, s);
          }
        }
        const uint32_t n_need = static_cast<uint32_t>(__popcll(need));
        const uint32_t room = 64u - tpos;
        const uint32_t taken = n_need < room ? n_need : room;
        tpos += taken;
        // the `taken` lowest needy lanes are served (even if their pixel fell outside the frame)
        need = __builtin_amdgcn_ballot_w64(needy && below >= taken);
      }
    }
    if (__builtin_amdgcn_ballot_w64(alive) == 0ull) {
      if (!work_left) break;
      continue;
    }
    wave_iters += 1u;
    {
      const bool was = alive;
      alive = path_step_wave<RT_DEFER_SKY_BLOCKS>(sv, fp, s, alive, nullptr, nullptr RT_PH_ARGS);
      if (was & !alive) {
        path_end<RT_DEFER_SKY_BLOCKS>(fp, s);
        seg_total += s.seg;
        rough_total += s.rough & 0x7fffffffu;
      }
    }
  }
  add_totals(fp, seg_total, rough_total, wave_iters);
}

#endif  // RT_AB_SHAPES

// Primary-ray culling pre-pass.  The rays of one pixel column lie in a plane through the camera
// (pixel_world_space is affine in v, ray.hlsl:31-35), and so do the rays of one pixel row, so a tile's
// frustum is the intersection of a column slab (between two such planes) and a row slab, and the set
// of spheres that can touch it factorises: mask(tile) = column_mask[tx] & row_mask[ty].  One wave per
// slab, lane k tests sphere 64*w + k; slabs are grown by one pixel on either side (the jitter is below
// half a pixel, connection.rs:20) and a sphere is dropped only if it lies entirely outside a slab even
// after its radius is inflated by 0.1 % + 0.2 % of its distance to the camera.  That margin exceeds
// the rounding error with which the float test of ray.hlsl:53-68 can still report a hit for a ray
// that geometrically misses (|q_float - q| <= 16 * 2^-24 * (D^2 + r^2): an effective radius error
// below 0.1 % of the distance D).  masks = [tiles_x column masks][tiles_y row masks], `mask_words`
// 64-bit words each.
__global__ void __launch_bounds__(256) cull_slabs_kernel(DeviceScene sc, FrameParams fp, unsigned long long *masks,
                                                         uint32_t tiles_x, uint32_t tiles_y) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t slab = blockIdx.x * 4u + (threadIdx.x >> 6);
  if (slab >= tiles_x + tiles_y) return;
  auto corner = [&](float u, float v) {  // ray direction (not normalised) of normalised pixel (u, v)
    f3 p = fma3(fp.cam_right, u * fp.vp_width, fp.top_left);
    p = fma3(fp.cam_neg_up, v * fp.vp_height, p);
    return p - fp.cam_pos;
  };
  f3 n_lo, n_hi;  // inward unit normals of the slab's two bounding planes
  if (slab < tiles_x) {
    const float px0 = static_cast<float>(slab * kTile) - 1.0f;
    const float px1 = static_cast<float>(min(slab * kTile + kTile, fp.width)) + 1.0f;
    const float u0 = px0 / static_cast<float>(fp.width), u1 = px1 / static_cast<float>(fp.width);
    const f3 a0 = corner(u0, 0.0f), b0 = corner(u0, 1.0f), a1 = corner(u1, 0.0f), b1 = corner(u1, 1.0f);
    const f3 inside = (a0 + b0) + (a1 + b1);
    n_lo = normalize3(cross3(a0, b0));
    n_hi = normalize3(cross3(a1, b1));
    if (dot3(n_lo, inside) < 0.0f) n_lo = -n_lo;
    if (dot3(n_hi, inside) < 0.0f) n_hi = -n_hi;
  } else {
    const uint32_t ty = slab - tiles_x;
    // rows of this tile row in full-frame coordinates (local rows follow the strip partition)
    const uint32_t lr0 = ty * kTile, lr1 = min(lr0 + kTile - 1u, fp.local_rows - 1u);
    const uint32_t y0 = ((lr0 / fp.strip_rows) * fp.world + fp.rank) * fp.strip_rows + lr0 % fp.strip_rows;
    const uint32_t y1 = ((lr1 / fp.strip_rows) * fp.world + fp.rank) * fp.strip_rows + lr1 % fp.strip_rows;
    const float v0 = (static_cast<float>(y0) - 1.0f) / static_cast<float>(fp.height);
    const float v1 = (static_cast<float>(y1 + 1u) + 1.0f) / static_cast<float>(fp.height);
    const f3 a0 = corner(0.0f, v0), b0 = corner(1.0f, v0), a1 = corner(0.0f, v1), b1 = corner(1.0f, v1);
    const f3 inside = (a0 + b0) + (a1 + b1);
    n_lo = normalize3(cross3(a0, b0));
    n_hi = normalize3(cross3(a1, b1));
    if (dot3(n_lo, inside) < 0.0f) n_lo = -n_lo;
    if (dot3(n_hi, inside) < 0.0f) n_hi = -n_hi;
  }
  const uint32_t words = fp.mask_words;
  for (uint32_t w = 0; w < words; ++w) {
    const uint32_t i = w * 64u + lane;
    bool keep = false;
    if (i < sc.n_sph) {
      const float4 g = sc.image[i];
      const f3 rel = mk3(g.x, g.y, g.z) - fp.cam_pos;
      const float dist = length3(rel);
      const float r = sqrtf(fmaxf(g.w, 0.0f)) * 1.001f + 0.002f * dist + 1e-6f;
      const float dlo = dot3(n_lo, rel), dhi = dot3(n_hi, rel);
      keep = (dlo >= -r) & (dhi >= -r);
      keep |= !(dlo == dlo) | !(dhi == dhi) | !(r == r);  // NaN anywhere: keep
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
    if (lane == 0) masks[static_cast<size_t>(slab) * words + w] = m;
  }
}

// Fills the direction-mask table (rt_dirmask.h): one thread per (cell, bin) entry.
__global__ void __launch_bounds__(256) dirmask_build_kernel(DirMaskParams dm, const float4 *reg, uint32_t n, unsigned long long *table,
                                                            unsigned long long entries) {
  __shared__ float4 s_reg[kDirMaskMaxSpheres];
  if (threadIdx.x < n) s_reg[threadIdx.x] = reg[threadIdx.x];  // n <= 128 < 256 threads
  __syncthreads();
  const unsigned long long e = static_cast<unsigned long long>(blockIdx.x) * 256ull + threadIdx.x;
  if (e > entries + 1ull) return;
  if (e >= entries) {  // the two extra entries: all-zero (lanes without a ray), the outliers alone (rays that miss the box)
    table[e * dm.words] = e == entries ? 0ull : dm.outlier_mask[0];
    if (dm.words > 1u) table[e * dm.words + 1u] = e == entries ? 0ull : dm.outlier_mask[1];
    return;
  }
  unsigned long long w[2];
  dirmask_entry(dm, static_cast<uint32_t>(e / dm.bins), static_cast<uint32_t>(e % dm.bins), s_reg, n, w);
  table[e * dm.words] = w[0];
  if (dm.words > 1u) table[e * dm.words + 1u] = w[1];
}

// Fills the candidate-list table of a large scene (rt_dirmask.h "Candidate lists"): one workgroup per cell.  It first collects,
// in slot order, the spheres near enough to be listed by ANY entry of the cell (lower distance bound within the largest reach) —
// a few dozen of a thousand — and every thread then builds the entries of its bins from those alone.  A cell with more near
// spheres than the LDS list holds is counted in *overflow and left without entries (reach 0: its rays scan linearly); the host
// then rebuilds with a shorter reach.  The block after the last cell writes the all-zero entry lanes without a lookup read.
constexpr uint32_t kDirListNearCap = 1024;
__global__ void __launch_bounds__(256) dirlist_build_kernel(DirMaskParams dm, const float4 *reg, uint32_t n, uint4 *table, uint32_t n_cells,
                                                            uint32_t *overflow) {
  __shared__ float4 near_rec[kDirListNearCap];
  __shared__ uint16_t near_slot[kDirListNearCap];
  __shared__ uint32_t wave_count[4];
  const uint32_t cell = blockIdx.x;
  if (cell >= n_cells) {
    if (threadIdx.x < 2u) table[2u * static_cast<size_t>(dm.entries) + threadIdx.x] = make_uint4(0u, 0u, 0u, 0u);
    return;
  }
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const DirListGeom g0 = dirlist_geometry(dm, cell, 0u);  // centre and extents are the cell's, whatever the bin
  const float reach_max = dm.reach_unit * static_cast<float>(kDirListReachQuanta);
  uint32_t count = 0;  // the same in every thread
  for (uint32_t base = 0; base < n; base += 256u) {
    const uint32_t i = base + threadIdx.x;
    bool near = false;
    float4 r = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (i < n) {
      r = reg[i];
      // dirlist_candidate's own distance bound: a sphere it can list has dmin <= reach_max (or touches the cell, or is NaN)
      const float vx = r.x - g0.p.x, vy = r.y - g0.p.y, vz = r.z - g0.p.z;
      const float l2 = vx * vx + vy * vy + vz * vz;
      const float R = r.w + g0.h;
      near = r.w < 0.0f || !(l2 > R * R * 1.0001f) || !(fmaxf(sqrtf(l2) * 0.9999f - R, 0.0f) > reach_max);  // (r.w < 0: an outlier, in every entry)
    }
    const unsigned long long m = __builtin_amdgcn_ballot_w64(near);
    if (lane == 0) wave_count[wave] = static_cast<uint32_t>(__popcll(m));
    __syncthreads();
    uint32_t before = count, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < 4u; ++w) {
      before += w < wave ? wave_count[w] : 0u;
      total += wave_count[w];
    }
    const uint32_t pos = before + __builtin_amdgcn_mbcnt_hi(static_cast<uint32_t>(m >> 32), __builtin_amdgcn_mbcnt_lo(static_cast<uint32_t>(m), 0u));
    if (near && pos < kDirListNearCap) {
      near_rec[pos] = r;
      near_slot[pos] = static_cast<uint16_t>(i);
    }
    count += total;
    __syncthreads();
  }
  const bool too_many = count > kDirListNearCap;
  if (too_many && threadIdx.x == 0) atomicAdd(overflow, 1u);
  for (uint32_t bin = threadIdx.x; bin < dm.bins; bin += 256u) {
    uint32_t w[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    if (!too_many) dirlist_entry(dm, cell, bin, near_rec, near_slot, count, w);
    uint4 *e = table + 2u * (static_cast<size_t>(cell) * dm.bins + bin);
    e[0] = make_uint4(w[0], w[1], w[2], w[3]);
    e[1] = make_uint4(w[4], w[5], w[6], w[7]);
  }
}

// Second pass of a batch launch (rt_render_frames): folds the batch's float samples into the history
// in frame order, i.e. replays frag.hlsl:60-65 + the Rgba8UnormSrgb store once per frame exactly as
// path_end would have done in `batch` separate launches (same expression, same quantisation of every
// intermediate frame), and stores the state after the last frame.
__global__ void __launch_bounds__(256) resolve_batch_kernel(ResolveParams rp) {
  const uint32_t p = blockIdx.x * 256u + threadIdx.x;
  if (p >= rp.pixels) return;
  f3 prev = mk3(0.0f, 0.0f, 0.0f);
  if (rp.count[0] != 0u) {
    if (rp.accum_mode == 1) {
      prev = mk3(rp.hist_f32[3 * static_cast<size_t>(p)], rp.hist_f32[3 * static_cast<size_t>(p) + 1],
                 rp.hist_f32[3 * static_cast<size_t>(p) + 2]);
    } else {
      const uint32_t w = rp.hist_rgba8[p];
      prev = mk3(rp.srgb_lut[w & 255u], rp.srgb_lut[(w >> 8) & 255u], rp.srgb_lut[(w >> 16) & 255u]);
    }
  }
  f3 mixed = prev;
  uint32_t texel = 0;
  for (uint32_t f = 0; f < rp.batch; ++f) {
    const size_t q = 3 * (static_cast<size_t>(f) * rp.frame_stride + p);
    const f3 sample = mk3(rp.samples[q], rp.samples[q + 1], rp.samples[q + 2]);
    if (rp.count[f] == 0u) prev = mk3(0.0f, 0.0f, 0.0f);  // path_end does not read the history then
    const float opacity = rp.opacity[f];
    const float keep = 1.0f - opacity;
    mixed = fma3(sample, opacity, prev * keep);
    texel = srgb8_encode_spec(mixed.x) | (srgb8_encode_spec(mixed.y) << 8) | (srgb8_encode_spec(mixed.z) << 16) |
            0xff000000u;
    prev = rp.accum_mode == 1 ? mixed
                              : mk3(rp.srgb_lut[texel & 255u], rp.srgb_lut[(texel >> 8) & 255u],
                                    rp.srgb_lut[(texel >> 16) & 255u]);
    if (rp.kept != nullptr && (f + 1u) % rp.keep_every == 0u)  // what the reference would have presented after frame f
      rp.kept[static_cast<size_t>(f / rp.keep_every) * rp.pixels + p] = texel;
  }
  rp.rgba8[p] = texel;
  if (rp.rgbf32) {
    rp.rgbf32[3 * static_cast<size_t>(p)] = mixed.x;
    rp.rgbf32[3 * static_cast<size_t>(p) + 1] = mixed.y;
    rp.rgbf32[3 * static_cast<size_t>(p) + 2] = mixed.z;
  }
  if (rp.hist_f32 && rp.accum_mode == 1) {
    rp.hist_f32[3 * static_cast<size_t>(p)] = mixed.x;
    rp.hist_f32[3 * static_cast<size_t>(p) + 1] = mixed.y;
    rp.hist_f32[3 * static_cast<size_t>(p) + 2] = mixed.z;
  }
}

// Rank 0 of a tile-split frame: frame row y belongs to strip y / strip_rows, dealt round-robin over the ranks
// (rt_set_partition); rank r keeps its strips as consecutive local rows.
__global__ void __launch_bounds__(256) assemble_strips_kernel(AssembleParams ap) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= ap.width * ap.height) return;
  const uint32_t y = i / ap.width, x = i - y * ap.width;
  const uint32_t strip = y / ap.strip_rows, r = strip % ap.world;
  const uint32_t local_row = (strip / ap.world) * ap.strip_rows + (y - strip * ap.strip_rows);
  const size_t src = static_cast<size_t>(local_row) * ap.width + x;
  ap.frame[i] = r == 0u ? ap.own[src] : ap.stack[ap.stack_offset[r] + src];
}

// Snapshot of a frame's texels for the asynchronous presentation (rt_present_async): a plain 16-bytes-per-lane copy on the stream
// that produced the frame — microseconds at HBM rate, where a hipMemcpyAsync between device buffers may take the DMA engine the
// host copies run on.
__global__ void __launch_bounds__(256) copy_texels_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16,
                                                          const unsigned char *__restrict__ src_tail, unsigned char *__restrict__ dst_tail,
                                                          uint32_t n_tail) {
  const size_t stride = static_cast<size_t>(gridDim.x) * 256u;
  for (size_t i = static_cast<size_t>(blockIdx.x) * 256u + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
  if (blockIdx.x == 0 && threadIdx.x < n_tail) dst_tail[threadIdx.x] = src_tail[threadIdx.x];
}

// Device-side refill of the random texture: the reference regenerates 600x600
// uniform floats on a host thread every 4 ms and uploads them (random_texture.rs:56-75, ~360 MB/s of
// PCIe); here texel k (row-major) is SplitMix64 output k+1 of `seed`, top 24 bits -> [0,1) on the
// 2^-24 grid — the same stream rsraytracingv3_amd.synth.random_texture produces on the host.
__global__ void random_fill_kernel(float *dst, uint32_t count, unsigned long long seed) {
  const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= count) return;
  unsigned long long z = seed + (static_cast<unsigned long long>(k) + 1ull) * 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  dst[k] = static_cast<float>(static_cast<uint32_t>(z >> 40)) * 5.9604644775390625e-8f;  // * 2^-24
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
    case 6: r = sqrtf(x[i]); break;
    default: r = sqrt_rn(x[i]); break;
  }
  y[i] = r;
}

__global__ void math_binary_kernel(int op, const float *a, const float *b, float *y, size_t n) {
  const size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
  if (i >= n) return;
  // ops 3 / 4: the sphere test's hit key bits(-dn - root(q)) - bits(EPSILON) for a = dn, b = q, with the scan's branch-free
  // root and with the IEEE root — equal for every (dn, q) the scan can ask about (tests/test_parity_gpu.py)
  if (op == 3) {
    y[i] = __uint_as_float(__float_as_uint(-a[i] - sqrt_for_hit(b[i])) - kEpsBits);
  } else if (op == 4) {
    y[i] = __uint_as_float(__float_as_uint(-a[i] - sqrtf(b[i])) - kEpsBits);
  } else if (op == 5) {  // the kernels' texel addressing: a = u, b = texture size (as a float)
    y[i] = static_cast<float>(texel_index(a[i], b[i], b[i] - 1.0f));
  } else {
    y[i] = op == 0 ? atan2_spec(a[i], b[i]) : (op == 1 ? pow_spec(a[i], b[i]) : a[i] / b[i]);
  }
}

// Every non-negative float through the kernel's sqrt_rn against hipcc's correctly rounded sqrtf.
__global__ void sqrt_exhaustive_kernel(unsigned long long *mismatches) {
  const uint32_t stride = gridDim.x * blockDim.x;
  uint32_t bad = 0;
  for (uint64_t u = static_cast<uint64_t>(blockIdx.x) * blockDim.x + threadIdx.x; u <= 0x7f800000ull; u += stride) {
    const float x = __uint_as_float(static_cast<uint32_t>(u));
    bad += __float_as_uint(sqrt_rn(x)) != __float_as_uint(sqrtf(x));
    {  // normalize3's reciprocal: the IEEE quotient for every float up to 2^126 in magnitude (both signs) and for inf / NaN
      const uint32_t mag = static_cast<uint32_t>(u) & 0x7fffffffu;
      if (mag <= 0x7e800000u || mag >= 0x7f800000u) {
        for (uint32_t sign = 0; sign < 2u; ++sign) {
          const float y = __uint_as_float(static_cast<uint32_t>(u) | (sign << 31));
          const uint32_t got = __float_as_uint(rcp_of_root(y)), want = __float_as_uint(1.0f / y);
          bad += got != want && !((got & 0x7fffffffu) > 0x7f800000u && (want & 0x7fffffffu) > 0x7f800000u);
        }
      }
    }
  }
  if (bad) atomicAdd(mismatches, static_cast<unsigned long long>(bad));
}

}  // namespace

namespace {
// sample-only launches (the staging buffer is the only output): multi-frame launches and pipelined single frames
bool sample_only(const FrameParams &fp) {
  return fp.rgba8 == nullptr && fp.rgbf32 != nullptr && fp.hist_f32 == nullptr && fp.segments == nullptr && fp.hits == nullptr &&
         fp.progressive_count == 0u && fp.opacity == 1.0f && fp.keep == 0.0f && !fp.cull_check;
}
uint64_t launch_tiles(const FrameParams &fp) {
  const uint32_t tiles_x = (fp.width + kTile - 1) / kTile;
  const uint32_t tiles_y = (fp.local_rows + kTile - 1) / kTile;
  return static_cast<uint64_t>(tiles_x) * tiles_y * fp.batch;  // the batch as one stacked image
}
}  // namespace

uint32_t stream_queue_shard_cap(const FrameParams &fp) {
#ifndef RT_AB_SHAPES
  (void)fp;
  return 0u;  // (the streamed shape lives in the A/B build only)
#else
  const uint32_t tiles_x = (fp.width + kTile - 1) / kTile, tiles_y = (fp.local_rows + kTile - 1) / kTile;
  const uint64_t blocks = static_cast<uint64_t>((tiles_x + RT_STREAM_NW1 - 1) / RT_STREAM_NW1) * tiles_y * fp.batch;  // primary_stream's grid
  return static_cast<uint32_t>((blocks + kQueueShards - 1) / kQueueShards) * RT_STREAM_NW1 * 64u;
#endif
}

int pick_variant(const DeviceScene &sc, const FrameParams &fp, const LaunchConfig &lc) {
  const uint64_t tiles = launch_tiles(fp);
  const size_t lds_sph = static_cast<size_t>(sc.n_sph_pad) * sizeof(float4);
  const bool image_fits = sc.image_records <= kImageLdsRecords;  // whole scene image in LDS (<= 16 KB)
  int variant = lc.variant;
  if (variant < 0) {
    // Measured on MI355X (tools/variant_sweep.py): the re-packing shape wins every multi-frame
    // launch down to a rank's eighth of a 1080p frame, and single-frame launches from about three quarters of a 1080p frame up;
    // scenes with hundreds of spheres (grid or not) and small single frames are fastest as plain tiles.  The
    // resident-wave shape (2) is never the fastest since the primary-ray culling, which it cannot use; selectable.
    // The streamed shape (4) only draws level from 16 frames per launch up; selectable.
    // single frames: without a direction-mask table a whole 1080p frame (32 400 tiles) re-packs, half of one renders as tiles;
    // with the table (whose balanced scan only the re-packing shape has) it wins down to a rank's eighth of a 1080p frame
    // (tools/variant_sweep.py, ms per frame tiles / re-packing: 1080p N = 2: 0.0737 / 0.0667, N = 4: 0.0420 / 0.0400, N = 8:
    // 0.0312 / 0.0309; 4K N = 8: 0.0683 / 0.0640)
    const uint64_t enough = fp.batch > 1u ? 20000ull : (sc.dm.enabled ? 4000ull : 24000ull);
    // up to 128 spheres (+ 8 padding records), or a large scene with candidate lists
    variant = (((sc.n_sph_pad > 136u || sc.grid.enabled) && !sc.dm.lists) || tiles < enough) ? 0 : 3;
#if defined(RT_AB_SHAPES) && RT_STREAM_MIN_TILES
    if (variant == 3 && tiles >= RT_STREAM_MIN_TILES && sample_only(fp)) variant = 4;
#endif
  }
#ifndef RT_AB_SHAPES
  if (variant == 2 || variant == 4) variant = 3;  // (the resident-wave and streamed shapes live in the A/B build only)
#endif
  // the streamed shape writes float samples only and needs its queue
  if (variant == 4 && !sample_only(fp)) variant = 3;
  // the packed path state of the streamed shape (and of the re-packing shape when it packs after the shading) keeps 6 bits of
  // bounce count and 26 bits of pixel index; the default re-packing shape carries both in full
  if (variant == 4 && (fp.reflection_limit > 63u || tiles * (kTile * kTile) > (1ull << 26))) variant = 3;
  // the streamed and resident-wave shapes have no candidate-list form, nor have the tiles that read everything from memory
  if ((variant == 2 || variant == 4) && sc.dm.lists) variant = 3;
  if (variant == 1 && sc.dm.lists) variant = 0;
  // the re-packing, streamed and resident-wave shapes keep the whole scene image in LDS; bigger scenes render as tiles (the re-packing
  // shape of a scene with candidate lists stages nothing)
  if ((variant == 3 || variant == 2 || variant == 4) && !image_fits && !sc.dm.lists) variant = 0;
  // tiles whose sphere records do not fit a workgroup's default 64 KB of LDS scan them from global memory
  if (variant == 0 && !image_fits && lds_sph > 60u * 1024u && !sc.dm.lists) variant = 1;
  return variant;
}

int launch_render(const DeviceScene &sc, const FrameParams &fp, const LaunchConfig &lc, hipStream_t stream, int *variant_used,
                  int *stage_used) {
  const uint32_t tiles_x = (fp.width + kTile - 1) / kTile;
  const uint32_t tiles_y = (fp.local_rows + kTile - 1) / kTile;
  if (fp.batch < 1u || fp.batch > kMaxBatch || fp.frame_tile_rows != tiles_y) return static_cast<int>(hipErrorInvalidValue);
  const uint64_t tiles = static_cast<uint64_t>(tiles_x) * tiles_y * fp.batch;  // the batch as one stacked image
  if (tiles == 0) return 0;
  if (tiles * (kTile * kTile) > 0xffffffffull) return static_cast<int>(hipErrorInvalidValue);  // pixel indices are 32-bit
  const size_t lds_sph = static_cast<size_t>(sc.n_sph_pad) * sizeof(float4);
  const size_t lds_image = static_cast<size_t>(sc.image_records) * sizeof(float4);
  const bool image_fits = sc.image_records <= kImageLdsRecords;
  int variant = pick_variant(sc, fp, lc);
  const bool dry = (lc.flags & 16) != 0;
  if (variant == 4 && fp.queue == nullptr) variant = 3;  // no queue handed in: the re-packing shape serves
  if (variant_used) *variant_used = variant;
  if (stage_used) *stage_used = 2;  // the re-packing and resident-wave shapes stage the whole image
#ifdef RT_AB_SHAPES
  if (variant == 2) {
    uint32_t blocks = static_cast<uint32_t>(lc.persistent_blocks);
    const uint64_t max_useful = (tiles + (kBlock / 64) - 1) / (kBlock / 64);
    if (blocks > max_useful) blocks = static_cast<uint32_t>(max_useful);
    if (blocks == 0) blocks = 1;
    const uint32_t chunk = static_cast<uint32_t>(lc.tile_chunk < 1 ? 1 : lc.tile_chunk);
    hipLaunchKernelGGL(render_persistent, dim3(blocks), dim3(kBlock), lds_image, stream, sc, fp,
                       static_cast<uint32_t>(lc.refill_min), chunk);
  } else if (variant == 4) {
    if (stage_used) *stage_used = 40;
    hipLaunchKernelGGL((primary_stream<RT_STREAM_NW1>), dim3((tiles_x + RT_STREAM_NW1 - 1) / RT_STREAM_NW1, tiles_y, fp.batch),
                       dim3(RT_STREAM_NW1 * 64), lds_image, stream, sc, fp);
    uint32_t blocks2 = static_cast<uint32_t>(lc.stream_blocks < 1 ? 1 : lc.stream_blocks);
    const uint32_t chunk = 64u * static_cast<uint32_t>(lc.tile_chunk < 1 ? 1 : lc.tile_chunk);
    hipLaunchKernelGGL((bounce_stream<RT_STREAM_NW2, RT_STREAM_WAVES>), dim3(blocks2), dim3(RT_STREAM_NW2 * 64), lds_image, stream, sc,
                       fp, static_cast<uint32_t>(lc.refill_min), chunk);
  } else
#endif
  if (variant == 3) {
    // (RT_OPT_FLAGS bit 4, "dry" launches: one workgroup per kernel — all of a launch's host work, next to none of its rendering)
    const bool lists = sc.dm.enabled && sc.dm.lists;
    const uint32_t nw = lists ? RT_BLOCK_NW_LISTS : RT_BLOCK_NW;
    const dim3 blocks = dry ? dim3(1, 1, 1) : dim3((tiles_x + nw - 1) / nw, tiles_y, fp.batch);  // x: tile columns, y: tile rows, z: frames
    // the staged scene image + exchange buffer and the waves' scan scratch; with candidate lists nothing is staged
    const size_t bytes = lists ? kBlocksLdsExtraLists : lds_image + kBlocksLdsExtra;
    const size_t tail_bytes = lists ? RT_TAIL_NW * kListScratchBytes : lds_image + RT_TAIL_NW * kScanScratchBytes;
    const bool lean = sample_only(fp);
    const bool spill = lean && fp.spill_max != 0u && fp.queue != nullptr;
    const dim3 threads(nw * 64);
#define RT_LAUNCH_BLOCKS(WAVES, WAVES_LISTS, LEAN_, SPILL_)                                                                                     \
  do {                                                                                                                                         \
    if (lists) hipLaunchKernelGGL((render_blocks<RT_BLOCK_NW_LISTS, WAVES_LISTS, LEAN_, SPILL_, true>), blocks, threads, bytes, stream, sc, fp); \
    else hipLaunchKernelGGL((render_blocks<RT_BLOCK_NW, WAVES, LEAN_, SPILL_, false>), blocks, threads, bytes, stream, sc, fp);                \
  } while (0)
    if (fp.batch > 1u && lean) {
      if (lists) {
        if (stage_used) *stage_used = 10 * RT_BLOCK_NW_LISTS + RT_BLOCKS_WAVES_LISTS;
        if (spill) RT_LAUNCH_BLOCKS(RT_BLOCKS_WAVES_BATCH, RT_BLOCKS_WAVES_LISTS, true, true);
        else RT_LAUNCH_BLOCKS(RT_BLOCKS_WAVES_BATCH, RT_BLOCKS_WAVES_LISTS, true, false);
      } else if (sc.dm.enabled && sc.dm.words > 1u) {
        if (stage_used) *stage_used = 10 * RT_BLOCK_NW + RT_BLOCKS_WAVES_BATCH_WIDE;
        if (spill) RT_LAUNCH_BLOCKS(RT_BLOCKS_WAVES_BATCH_WIDE, RT_BLOCKS_WAVES_LISTS, true, true);
        else RT_LAUNCH_BLOCKS(RT_BLOCKS_WAVES_BATCH_WIDE, RT_BLOCKS_WAVES_LISTS, true, false);
      } else {
        if (stage_used) *stage_used = 10 * RT_BLOCK_NW + RT_BLOCKS_WAVES_BATCH;
        if (spill) RT_LAUNCH_BLOCKS(RT_BLOCKS_WAVES_BATCH, RT_BLOCKS_WAVES_LISTS, true, true);
        else RT_LAUNCH_BLOCKS(RT_BLOCKS_WAVES_BATCH, RT_BLOCKS_WAVES_LISTS, true, false);
      }
    } else if (lean) {
      if (stage_used) *stage_used = 10 * static_cast<int>(nw) + (lists ? RT_BLOCKS_WAVES_SINGLE_LISTS : RT_BLOCKS_WAVES_SINGLE);
      if (spill) RT_LAUNCH_BLOCKS(RT_BLOCKS_WAVES_SINGLE, RT_BLOCKS_WAVES_SINGLE_LISTS, true, true);
      else RT_LAUNCH_BLOCKS(RT_BLOCKS_WAVES_SINGLE, RT_BLOCKS_WAVES_SINGLE_LISTS, true, false);
    } else {
      if (stage_used) *stage_used = 10 * static_cast<int>(nw) + RT_BLOCKS_WAVES_FULL;
      RT_LAUNCH_BLOCKS(RT_BLOCKS_WAVES_FULL, RT_BLOCKS_WAVES_FULL, false, false);
    }
#undef RT_LAUNCH_BLOCKS
    if (spill) {  // the spilled paths' remaining bounces, in full waves
      if (stage_used) *stage_used += 100;
      if (lc.tail_start != nullptr) (void)hipEventRecord(lc.tail_start, stream);
      const uint32_t blocks2 = dry ? 1u : static_cast<uint32_t>(lc.tail_blocks < 1 ? 1 : lc.tail_blocks);
      const uint32_t chunk = 64u * static_cast<uint32_t>(lc.tile_chunk < 1 ? 1 : lc.tile_chunk);
      if (lists)
        hipLaunchKernelGGL((tail_stream<RT_TAIL_NW, RT_TAIL_WAVES_LISTS, true>), dim3(blocks2), dim3(RT_TAIL_NW * 64), tail_bytes, stream, sc, fp,
                           static_cast<uint32_t>(lc.refill_min), chunk);
      else
        hipLaunchKernelGGL((tail_stream<RT_TAIL_NW, RT_TAIL_WAVES, false>), dim3(blocks2), dim3(RT_TAIL_NW * 64), tail_bytes, stream, sc, fp,
                           static_cast<uint32_t>(lc.refill_min), chunk);
    }
  } else {
    const dim3 blocks = dry ? dim3(1, 1, 1) : dim3((tiles_x + (kTileBlock / 64) - 1) / (kTileBlock / 64), tiles_y, fp.batch);
    const bool lists = sc.dm.enabled && sc.dm.lists;
    if (stage_used) *stage_used = variant == 1 ? 0 : (lists ? 4 : ((sc.grid.enabled && !(lc.flags & 2)) ? 3 : (image_fits ? 2 : 1)));
    if (variant == 1) {
      hipLaunchKernelGGL(render_tiles<0>, blocks, dim3(kTileBlock), 0, stream, sc, fp);
    } else if (lists) {
      hipLaunchKernelGGL(render_tiles<4>, blocks, dim3(kTileBlock), (kTileBlock / 64) * kListScratchBytes, stream, sc, fp);
    } else if (sc.grid.enabled && !(lc.flags & 2)) {  // RT_OPT_FLAGS bit 1: ignore the grid (A/B)
      const size_t bytes = lds_sph + (static_cast<size_t>(sc.grid.off_items - sc.grid.off_cells) +
                                      (static_cast<size_t>(sc.grid.n_items) * 2 + 15) / 16) * sizeof(float4);
      if (bytes <= 12u * 1024u) {
        hipLaunchKernelGGL((render_tiles<3, 128>), dry ? dim3(1, 1, 1) : dim3((tiles_x + 1) / 2, tiles_y, fp.batch), dim3(128), bytes, stream, sc, fp);
      } else {
        hipLaunchKernelGGL((render_tiles<3, kTileBlock>), blocks, dim3(kTileBlock), bytes, stream, sc, fp);
      }
    } else if (image_fits) {
      hipLaunchKernelGGL(render_tiles<2>, blocks, dim3(kTileBlock), lds_image, stream, sc, fp);
    } else {
      hipLaunchKernelGGL(render_tiles<1>, blocks, dim3(kTileBlock), lds_sph, stream, sc, fp);
    }
  }
  return static_cast<int>(hipGetLastError());
}

int launch_resolve(const ResolveParams &rp, hipStream_t stream, bool dry) {
  if (rp.pixels == 0 || rp.batch == 0) return 0;
  hipLaunchKernelGGL(resolve_batch_kernel, dim3(dry ? 1u : (rp.pixels + 255u) / 256u), dim3(256), 0, stream, rp);
  return static_cast<int>(hipGetLastError());
}

int launch_assemble(const AssembleParams &ap, hipStream_t stream, bool dry) {
  const uint64_t n = static_cast<uint64_t>(ap.width) * ap.height;
  if (n == 0) return 0;
  hipLaunchKernelGGL(assemble_strips_kernel, dim3(dry ? 1u : static_cast<uint32_t>((n + 255u) / 256u)), dim3(256), 0, stream, ap);
  return static_cast<int>(hipGetLastError());
}

int launch_cull_tiles(const DeviceScene &sc, const FrameParams &fp, unsigned long long *masks, hipStream_t stream) {
  const uint32_t tiles_x = (fp.width + kTile - 1) / kTile;
  const uint32_t tiles_y = (fp.local_rows + kTile - 1) / kTile;
  if (tiles_x * tiles_y == 0 || fp.mask_words == 0) return 0;
  hipLaunchKernelGGL(cull_slabs_kernel, dim3((tiles_x + tiles_y + 3) / 4), dim3(256), 0, stream, sc, fp, masks, tiles_x,
                     tiles_y);
  return static_cast<int>(hipGetLastError());
}

int launch_dirmask_build(const DirMaskParams &dm, const float4 *reg, uint32_t n, unsigned long long *table, hipStream_t stream) {
  const unsigned long long entries = static_cast<unsigned long long>(dm.nx) * dm.ny * dm.nz * dm.bins;
  if (entries == 0 || n == 0 || n > kDirMaskMaxSpheres) return static_cast<int>(hipErrorInvalidValue);
  hipLaunchKernelGGL(dirmask_build_kernel, dim3(static_cast<uint32_t>((entries + 257ull) / 256ull)), dim3(256), 0, stream, dm, reg, n,
                     table, entries);  // entries + 2 threads: the last two write the extra entries
  return static_cast<int>(hipGetLastError());
}

int launch_dirlist_build(const DirMaskParams &dm, const float4 *reg, uint32_t n, unsigned long long *table, uint32_t *d_overflow,
                         hipStream_t stream) {
  const unsigned long long cells = static_cast<unsigned long long>(dm.nx) * dm.ny * dm.nz;
  if (cells == 0 || cells > 0x7fffffffull || n == 0 || n > kDirListMaxSpheres || !dm.lists) return static_cast<int>(hipErrorInvalidValue);
  hipLaunchKernelGGL(dirlist_build_kernel, dim3(static_cast<uint32_t>(cells) + 1u), dim3(256), 0, stream, dm, reg, n,
                     reinterpret_cast<uint4 *>(table), static_cast<uint32_t>(cells), d_overflow);
  return static_cast<int>(hipGetLastError());
}

uint32_t tail_spill_plan(const DeviceScene &sc, const FrameParams &fp, const LaunchConfig &lc, uint32_t wanted, uint32_t *shard_cap) {
  if (shard_cap) *shard_cap = 0;
  // sample-only launches of the re-packing shape whose secondary rays use the balanced scan (tail_stream has its scratch)
  if (wanted == 0u || !sample_only(fp) || !sc.dm.enabled || pick_variant(sc, fp, lc) != 3) return 0u;
  // nothing to hand over when the first shading is the last; and tail_stream costs a launch ~35 us whatever it holds, which
  // small launches do not earn back (RT_OPT_FLAGS bit 2 lifts the size rule: tests)
  const uint64_t min_tiles = fp.batch > 1u ? ((sc.dm.enabled && sc.dm.lists) ? kTailMinTilesLists : kTailMinTiles) : kTailMinTilesSingle;
  if (fp.batch == 1u && !(lc.flags & 4) && fp.reflection_limit < ((sc.dm.enabled && sc.dm.lists) ? kTailMinBouncesSingleLists : kTailMinBouncesSingle))
    return 0u;
  if (fp.reflection_limit < 2u || (launch_tiles(fp) < min_tiles && !(lc.flags & 4))) return 0u;
  const uint32_t tiles_x = (fp.width + kTile - 1) / kTile, tiles_y = (fp.local_rows + kTile - 1) / kTile;
  const uint32_t nw = (sc.dm.enabled && sc.dm.lists) ? RT_BLOCK_NW_LISTS : RT_BLOCK_NW;
  const uint64_t blocks = static_cast<uint64_t>((tiles_x + nw - 1) / nw) * tiles_y * fp.batch;  // render_blocks' grid
  const uint64_t per_shard = (blocks + kQueueShards - 1) / kQueueShards;  // workgroups that append to one shard, each once
  uint64_t t = wanted > nw * 64u ? nw * 64u : wanted;
  const uint64_t fit = kTailQueueBudget / (per_shard * kQueueShards * 5ull * sizeof(float4));
  if (t > fit) t = fit;
  if (t < 4u || per_shard * t > 0xffffffffull) return 0u;
  if (shard_cap) *shard_cap = static_cast<uint32_t>(per_shard * t);
  return static_cast<uint32_t>(t);
}

int tail_blocks_per_cu(size_t lds_bytes, bool lists) {
  int n = 0;
  const hipError_t e = lists ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, tail_stream<RT_TAIL_NW, RT_TAIL_WAVES_LISTS, true>, RT_TAIL_NW * 64,
                                                                            RT_TAIL_NW * kListScratchBytes)
                             : hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, tail_stream<RT_TAIL_NW, RT_TAIL_WAVES, false>, RT_TAIL_NW * 64,
                                                                            lds_bytes + RT_TAIL_NW * kScanScratchBytes);
  return e == hipSuccess ? n : 0;
}

int stream_blocks_per_cu(size_t lds_bytes) {
#ifdef RT_AB_SHAPES
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, bounce_stream<RT_STREAM_NW2, RT_STREAM_WAVES>, RT_STREAM_NW2 * 64, lds_bytes) !=
      hipSuccess)
    return 0;
  return n;
#else
  (void)lds_bytes;
  return 0;
#endif
}

int prof_counts(unsigned long long out8[8], bool reset) {
#ifdef RT_PROFILE_EVENTS
  if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_prof_cnt), 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) {
    const unsigned long long zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_prof_cnt), zero, sizeof zero) != hipSuccess) return 1;
  }
  return 0;
#else
  for (int k = 0; k < 8; ++k) out8[k] = 0;
  return 0;
#endif
}

int persistent_blocks_per_cu(size_t lds_bytes) {
#ifdef RT_AB_SHAPES
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, render_persistent, kBlock, lds_bytes) != hipSuccess) return 0;
  return n;
#else
  (void)lds_bytes;
  return 0;
#endif
}

bool has_ab_shapes() {
#ifdef RT_AB_SHAPES
  return true;
#else
  return false;
#endif
}

int launch_random_fill(float *dst, uint32_t count, unsigned long long seed, hipStream_t stream) {
  if (count == 0) return 0;
  hipLaunchKernelGGL(random_fill_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, dst, count, seed);
  return static_cast<int>(hipGetLastError());
}

int launch_copy_texels(const void *src, void *dst, size_t bytes, hipStream_t stream) {
  if (bytes == 0) return 0;
  const size_t n16 = bytes / 16u;
  const uint32_t tail = static_cast<uint32_t>(bytes - n16 * 16u);
  const size_t want = (n16 + 1023u) / 1024u;  // four 16-byte words per thread
  const uint32_t blocks = static_cast<uint32_t>(want < 1u ? 1u : (want > 4096u ? 4096u : want));
  hipLaunchKernelGGL(copy_texels_kernel, dim3(blocks), dim3(256), 0, stream, static_cast<const uint4 *>(src), static_cast<uint4 *>(dst), n16,
                     static_cast<const unsigned char *>(src) + n16 * 16u, static_cast<unsigned char *>(dst) + n16 * 16u, tail);
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

int launch_sqrt_exhaustive(unsigned long long *d_mismatches, hipStream_t stream) {
  hipLaunchKernelGGL(sqrt_exhaustive_kernel, dim3(4096), dim3(256), 0, stream, d_mismatches);
  return static_cast<int>(hipGetLastError());
}

}  // namespace rt355
