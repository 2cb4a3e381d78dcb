// rt_dirmask.h — candidate-sphere masks for secondary rays ("direction masks"), shared by host and device code.
//
// ray.hlsl:97-107 tests every object for every ray segment.  For scenes of up to 128 spheres (the reference's buffer
// holds 80 objects) rt355 keeps a table
//     mask[cell of the ray's origin][bin of the ray's direction] -> set of sphere slots (one or two 64-bit words)
// built once per sphere geometry (dirmask_build_kernel, rt_kernels.hip): a secondary ray looks its set up with one
// 8-byte load and tests only those spheres, in ascending slot order, with the arithmetic of the linear scan.  The
// result is that of the linear scan because (1) a sphere gives the same hit key wherever it is tested, (2) ascending
// slots with a strict '<' select the lexicographic minimum of (key, slot) — the reference's rule (App. B D10) — and
// (3) the set is CONSERVATIVE: it holds every sphere the float test can accept for any ray that maps to the entry.
//
// Why (3) holds.  As in rt_grid.h: for a ray that starts inside the near box (the spheres' bounds grown by a quarter of
// their diagonal; D_max = that box's diagonal) with a direction within 4e-6 of unit length, every (ray, sphere) pair the
// float test accepts has its accepted point x = o + t d, t > 0, inside the registered ball |x - c| <= r_reg,
//     r_reg = 1.001 * sqrt(r^2 + (0.003 * D_max)^2) + 1e-3 * cell edge.
// Cells: a cubic grid over the box B of all registered balls (grown by 1e-3 of its diagonal).  A ray that starts outside B
// is moved to the point where it enters B (every registered ball lies inside B, so nothing is skipped; a ray that misses
// B gets the empty set) and is then treated like a ray that starts there.
// Bins: a cube map of the direction, k x k bins per face: major axis by largest magnitude, (u, v) = the two other
// components divided by that magnitude.
// Entry (cell, bin): let p be the cell's centre, h its half diagonal (+ slack), a the unit vector through the bin's
// centre and theta the largest angle between a and any direction of the (slightly grown) bin.  If a ray from o in the
// cell with direction d in the bin reaches x with |x - c| <= r_reg, then |p + t d - c| <= r_reg + h =: R, so either
// |c - p| <= R, or the angle between d and c - p is at most asin(R / |c - p|), hence the angle between a and c - p is at
// most theta + asin(R / |c - p|).  A sphere is entered unless that fails with a margin; all slack terms (float
// rounding of the lane's cell and bin arithmetic, of the entry point, of the builder itself) are orders of magnitude
// below the margins used.  Rays outside the near box, with a non-unit direction or with a NaN anywhere do not use the
// table: their wave scans linearly.  RT_OPT_CULL = 2 re-scans every such lookup linearly inside the kernel and counts
// disagreements; tests/test_dirmask_host.py checks the sets against the oracle's rays on the CPU.
#pragma once

#include <cmath>
#include <cstdint>

#include "rt_math.h"

namespace rt355 {

constexpr uint32_t kDirMaskMaxSpheres = 128;  // one or two 64-bit words per entry
constexpr uint32_t kDirMaskMaxDim = 64;       // cells per axis
constexpr uint32_t kDirMaskMaxK = 16;         // bins per cube-map face edge

struct DirMaskParams {
  uint32_t enabled;
  uint32_t nx, ny, nz;          // cells per axis
  uint32_t k, bins;             // bins per face edge; 6 * k * k
  uint32_t words;               // 64-bit words per entry: 1 (up to 64 spheres) or 2
  uint32_t entries;             // nx * ny * nz * bins; the device table holds one more, all-zero entry at this index
  float half_k;                 // k / 2
  float bmin[3], bmax[3];       // the box B
  float cell[3], inv_cell[3];   // cell edge per axis (equal up to rounding: cubic cells) and its reciprocal
  float near_min[3], near_max[3];
  const unsigned long long *table;  // [cell][bin][words], device memory
};

// Bin of a direction (any non-zero vector): face = 2 * major axis + (major component < 0), (u, v) = the components
// (axis + 1) % 3 and (axis + 2) % 3 over the major magnitude, each cut into k steps of [-1, 1].
// `inv_major` = 1 / largest magnitude (the device passes v_rcp_f32's value; the builder's bins are grown to cover it).
RT_HD uint32_t dirmask_bin(f3 d, float inv_major, uint32_t k, float half_k) {
  const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
  float major, u, v;
  uint32_t face;
  if (ax >= ay && ax >= az) {
    major = d.x; u = d.y; v = d.z; face = 0u;
  } else if (ay >= az) {
    major = d.y; u = d.z; v = d.x; face = 2u;
  } else {
    major = d.z; u = d.x; v = d.y; face = 4u;
  }
  face += major < 0.0f ? 1u : 0u;
  // (u * inv + 1) * k / 2 in [0, k]; the conversion truncates (and sends negatives and NaN to 0), the last bin keeps k
  const float fu = fmaf(u * inv_major, half_k, half_k), fv = fmaf(v * inv_major, half_k, half_k);
  uint32_t bu = static_cast<uint32_t>(fmaxf(fu, 0.0f)), bv = static_cast<uint32_t>(fmaxf(fv, 0.0f));
  const uint32_t last = k - 1u;
  bu = bu < last ? bu : last;  // one v_min_u32 each
  bv = bv < last ? bv : last;
  return (face * k + bv) * k + bu;
}

// Cell of a point (clamped into the grid).
RT_HD uint32_t dirmask_cell(const DirMaskParams &dm, f3 p) {
  const float fx = (p.x - dm.bmin[0]) * dm.inv_cell[0], fy = (p.y - dm.bmin[1]) * dm.inv_cell[1],
              fz = (p.z - dm.bmin[2]) * dm.inv_cell[2];
  uint32_t ix = static_cast<uint32_t>(fmaxf(fx, 0.0f)), iy = static_cast<uint32_t>(fmaxf(fy, 0.0f)),
           iz = static_cast<uint32_t>(fmaxf(fz, 0.0f));
  const uint32_t lx = dm.nx - 1u, ly = dm.ny - 1u, lz = dm.nz - 1u;
  ix = ix < lx ? ix : lx;  // one v_min_u32 each
  iy = iy < ly ? iy : ly;
  iz = iz < lz ? iz : lz;
  return (iz * dm.ny + iy) * dm.nx + ix;
}

// One table entry: the set of spheres {c, r_reg} (reg[i] = {cx, cy, cz, r_reg}, n <= 128) a ray from cell `cell` with a
// direction of bin `bin` can be accepted by, as two 64-bit words (out[1] = 0 for n <= 64).  Plain float arithmetic with
// wide margins; host and device may round differently — both results are conservative.
RT_HD void dirmask_entry(const DirMaskParams &dm, uint32_t cell, uint32_t bin, const float4 *reg, uint32_t n, unsigned long long out[2]) {
  const uint32_t ix = cell % dm.nx, iy = (cell / dm.nx) % dm.ny, iz = cell / (dm.nx * dm.ny);
  const f3 p = mk3(dm.bmin[0] + (static_cast<float>(ix) + 0.5f) * dm.cell[0], dm.bmin[1] + (static_cast<float>(iy) + 0.5f) * dm.cell[1],
                   dm.bmin[2] + (static_cast<float>(iz) + 0.5f) * dm.cell[2]);
  const float diag = sqrtf(dm.cell[0] * dm.cell[0] + dm.cell[1] * dm.cell[1] + dm.cell[2] * dm.cell[2]);
  const float box = fmaxf(fmaxf(dm.bmax[0] - dm.bmin[0], dm.bmax[1] - dm.bmin[1]), dm.bmax[2] - dm.bmin[2]);
  const float h = 0.5f * diag * 1.01f + 2e-3f * box;  // half diagonal + slack for the entry point's and the cell index's rounding
  // the bin: centre direction and the largest angle to a corner of the bin grown by 2 % of its width + 1e-4
  const uint32_t k = dm.k;
  const uint32_t bu = bin % k, bv = (bin / k) % k, face = bin / (k * k);
  const uint32_t axis = face >> 1;
  const float sign = (face & 1u) ? -1.0f : 1.0f;
  const float step = 2.0f / static_cast<float>(k), grow = 0.02f * step + 1e-4f;
  const float u0 = -1.0f + step * static_cast<float>(bu) - grow, u1 = -1.0f + step * static_cast<float>(bu + 1u) + grow;
  const float v0 = -1.0f + step * static_cast<float>(bv) - grow, v1 = -1.0f + step * static_cast<float>(bv + 1u) + grow;
  auto dir = [&](float u, float v) {
    float c[3];
    c[axis] = sign;
    c[(axis + 1u) % 3u] = u;
    c[(axis + 2u) % 3u] = v;
    const float inv = 1.0f / sqrtf(c[0] * c[0] + c[1] * c[1] + c[2] * c[2]);
    return mk3(c[0] * inv, c[1] * inv, c[2] * inv);
  };
  const f3 a = dir(0.5f * (u0 + u1), 0.5f * (v0 + v1));
  float cos_theta = 1.0f;
  {
    const f3 c00 = dir(u0, v0), c01 = dir(u0, v1), c10 = dir(u1, v0), c11 = dir(u1, v1);
    cos_theta = fminf(fminf(a.x * c00.x + a.y * c00.y + a.z * c00.z, a.x * c01.x + a.y * c01.y + a.z * c01.z),
                      fminf(a.x * c10.x + a.y * c10.y + a.z * c10.z, a.x * c11.x + a.y * c11.y + a.z * c11.z));
  }
  // theta grown by 2e-3 rad: covers directions that are unit vectors only within 4e-6 and every rounding above
  cos_theta = fminf(cos_theta, 1.0f);
  const float sin_theta0 = sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
  const float ct = cos_theta * 0.999998f - sin_theta0 * 2e-3f, st = sin_theta0 * 0.999998f + cos_theta * 2e-3f;  // cos / sin of theta + 2e-3
  unsigned long long m0 = 0ull, m1 = 0ull;
  for (uint32_t i = 0; i < n; ++i) {
    const float4 g = reg[i];
    const float vx = g.x - p.x, vy = g.y - p.y, vz = g.z - p.z;
    const float l2 = vx * vx + vy * vy + vz * vz;
    const float R = g.w + h;
    bool in = !(l2 > R * R * 1.0001f);  // the cell (grown) touches the registered ball; NaN -> in
    if (!in) {
      const float inv_l = 1.0f / sqrtf(l2);
      const float s = fminf(R * inv_l * 1.0001f, 1.0f);  // sin(phi), phi = asin(R / |c - p|)
      const float c = sqrtf(fmaxf(1.0f - s * s, 0.0f));  // cos(phi)
      const float cos_sum = ct * c - st * s;             // cos(theta + phi) while theta + phi <= pi
      const float cos_ang = (a.x * vx + a.y * vy + a.z * vz) * inv_l;
      // theta + phi >= pi/2 + ... : cos_sum is decreasing in both on [0, pi]; theta < pi/2 and phi <= pi/2 keep the sum below pi
      in = !(cos_ang < cos_sum - 1e-4f);
    }
    if (i < 64u) m0 |= in ? (1ull << i) : 0ull;
    else m1 |= in ? (1ull << (i - 64u)) : 0ull;
  }
  out[0] = m0;
  out[1] = m1;
}

// Host side: resolution and boxes for n spheres {cx, cy, cz, r*r}; reg[i] receives {c, r_reg}.  Returns a disabled
// plan for scenes the table does not serve (no sphere, more than 128, non-finite data).  `target_cells`: cells aimed at
// (cubic cells, 1..kDirMaskMaxDim per axis); `k`: bins per face edge.
inline DirMaskParams dirmask_plan(const float4 *sph, uint32_t n, uint32_t target_cells, uint32_t k, float4 *reg) {
  DirMaskParams dm{};
  if (n == 0 || n > kDirMaskMaxSpheres || k < 1 || k > kDirMaskMaxK || target_cells < 1) return dm;
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  double r[kDirMaskMaxSpheres];
  for (uint32_t i = 0; i < n; ++i) {
    const double c[3] = {sph[i].x, sph[i].y, sph[i].z};
    if (!(sph[i].w >= 0.0f) || !std::isfinite(sph[i].w) || !std::isfinite(c[0]) || !std::isfinite(c[1]) || !std::isfinite(c[2])) return dm;
    r[i] = std::sqrt(static_cast<double>(sph[i].w));
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::fmin(lo[a], c[a] - r[i]);
      hi[a] = std::fmax(hi[a], c[a] + r[i]);
    }
  }
  const double diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
  if (!(diag > 0.0) || !std::isfinite(diag) || diag > 1e15) return dm;
  double near_lo[3], near_hi[3], near_diag = 0.0;
  for (int a = 0; a < 3; ++a) {
    near_lo[a] = lo[a] - 0.25 * diag;
    near_hi[a] = hi[a] + 0.25 * diag;
    near_diag += (near_hi[a] - near_lo[a]) * (near_hi[a] - near_lo[a]);
  }
  near_diag = std::sqrt(near_diag);
  const double eps = 0.003 * near_diag;
  // first pass without the cell term to size the cells, second with it
  double rr[kDirMaskMaxSpheres], glo[3], ghi[3], ext[3], edge = 0.0;
  uint32_t dims[3] = {1, 1, 1};
  for (int pass = 0; pass < 2; ++pass) {
    for (int a = 0; a < 3; ++a) {
      glo[a] = INFINITY;
      ghi[a] = -INFINITY;
    }
    for (uint32_t i = 0; i < n; ++i) {
      rr[i] = 1.001 * std::sqrt(r[i] * r[i] + eps * eps) + 1e-3 * edge;
      const double c[3] = {sph[i].x, sph[i].y, sph[i].z};
      for (int a = 0; a < 3; ++a) {
        glo[a] = std::fmin(glo[a], c[a] - rr[i]);
        ghi[a] = std::fmax(ghi[a], c[a] + rr[i]);
      }
    }
    double bdiag = 0.0;
    for (int a = 0; a < 3; ++a) bdiag += (ghi[a] - glo[a]) * (ghi[a] - glo[a]);
    bdiag = std::sqrt(bdiag);
    for (int a = 0; a < 3; ++a) {
      glo[a] -= 1e-3 * bdiag;
      ghi[a] += 1e-3 * bdiag;
      ext[a] = ghi[a] - glo[a];
    }
    edge = std::cbrt(ext[0] * ext[1] * ext[2] / static_cast<double>(target_cells));
    for (int a = 0; a < 3; ++a) {
      const double want = edge > 0.0 ? ext[a] / edge : 1.0;
      dims[a] = static_cast<uint32_t>(std::fmin(static_cast<double>(kDirMaskMaxDim), std::fmax(1.0, std::ceil(want - 1e-9))));
    }
    // cubic cells: the edge that makes every axis fit its cell count
    edge = 0.0;
    for (int a = 0; a < 3; ++a) edge = std::fmax(edge, ext[a] / dims[a]);
  }
  for (int a = 0; a < 3; ++a) {  // the grid may overhang the box on the high side (cells stay cubic)
    dm.bmin[a] = static_cast<float>(glo[a]);
    dm.bmax[a] = static_cast<float>(ghi[a]);
    dm.cell[a] = static_cast<float>(edge);
    dm.inv_cell[a] = static_cast<float>(1.0 / edge);
    dm.near_min[a] = static_cast<float>(near_lo[a] + 1e-6 * near_diag);
    dm.near_max[a] = static_cast<float>(near_hi[a] - 1e-6 * near_diag);
  }
  dm.nx = dims[0];
  dm.ny = dims[1];
  dm.nz = dims[2];
  dm.k = k;
  dm.bins = 6u * k * k;
  dm.words = n > 64u ? 2u : 1u;
  dm.entries = dm.nx * dm.ny * dm.nz * dm.bins;
  dm.half_k = 0.5f * static_cast<float>(k);
  for (uint32_t i = 0; i < n; ++i) reg[i] = make_float4(sph[i].x, sph[i].y, sph[i].z, static_cast<float>(rr[i] * 1.0001));
  dm.enabled = 1;
  dm.table = nullptr;
  return dm;
}

}  // namespace rt355
