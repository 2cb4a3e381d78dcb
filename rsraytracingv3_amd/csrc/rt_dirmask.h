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

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

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
  // candidate LISTS (scenes of more than kDirMaskMaxSpheres spheres; "Candidate lists" below): entries are 32 bytes
  // (words = 4), `table` holds them
  uint32_t lists;      // 1: list entries
  float reach_unit;    // world units per quantum of an entry's reach
  float s_cap;         // lookups are served for ray parameters up to this; beyond it a ray scans linearly
  // spheres far from all the others ("Outliers" below) do not size the table; they are members of every set / list instead
  uint32_t n_outliers;
  uint32_t outlier_slot[4];
  unsigned long long outlier_mask[2];  // their bits (bit sets: the extra entry rays that miss the box read)
};
constexpr uint32_t kDirMaskMaxOutliers = 4;

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
    bool in = g.w < 0.0f || !(l2 > R * R * 1.0001f);  // an outlier (registered with radius -1: in EVERY set), or the cell (grown) touches the registered ball; NaN -> in
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

// Outliers.  Everything above is sized by the spheres' bounding box: the cells divide the box of all registered balls, and the
// rounding term of every registered radius is 0.003 x the near box's diagonal.  ONE sphere far from the others (a distant light, an
// object dragged away in the editor) makes the cells ten times coarser and every ball fatter — the headline scene plus a sphere 860
// units away ran at 37 Grays/s instead of 57.  So up to four spheres far from all the others are taken out of the table's GEOMETRY: the
// boxes and the rounding term are those of the remaining spheres, and the outliers are members of EVERY set instead (no cone test:
// the builder lists them unconditionally) — also of the extra set behind the table that rays missing the box read —, so a set still
// holds every sphere the float test can accept.  (A ray that starts on a distant outlier is outside the near box and not served, as before.)
// Which: with c_med the per-axis median of the centres and e_i = |c_i - c_med| + r_i, the spheres with e_i > 3 x the 75th percentile
// of e, farthest first, while at least eight spheres remain.
inline uint32_t dirmask_outliers(const float4 *sph, uint32_t n, uint32_t out[kDirMaskMaxOutliers]) {
  if (n < 12u) return 0u;
  std::vector<double> v(n);
  double med[3];
  for (int a = 0; a < 3; ++a) {
    for (uint32_t i = 0; i < n; ++i) v[i] = a == 0 ? sph[i].x : (a == 1 ? sph[i].y : sph[i].z);
    std::nth_element(v.begin(), v.begin() + n / 2, v.end());
    med[a] = v[n / 2];
  }
  std::vector<double> e(n);
  for (uint32_t i = 0; i < n; ++i) {
    const double dx = sph[i].x - med[0], dy = sph[i].y - med[1], dz = sph[i].z - med[2];
    e[i] = std::sqrt(dx * dx + dy * dy + dz * dz) + std::sqrt(std::fmax(static_cast<double>(sph[i].w), 0.0));
  }
  v = e;
  std::nth_element(v.begin(), v.begin() + (3 * n) / 4, v.end());
  const double limit = 3.0 * v[(3 * n) / 4];
  if (!(limit > 0.0)) return 0u;
  uint32_t count = 0;
  for (; count < kDirMaskMaxOutliers; ++count) {
    uint32_t best = n;
    for (uint32_t i = 0; i < n; ++i) {
      bool taken = false;
      for (uint32_t j = 0; j < count; ++j) taken = taken || out[j] == i;
      if (!taken && e[i] > limit && (best == n || e[i] > e[best])) best = i;
    }
    if (best == n) break;
    out[count] = best;
  }
  std::sort(out, out + count);  // ascending slots
  return count;
}

// Host side: resolution and boxes for n spheres {cx, cy, cz, r*r}; reg[i] receives {c, r_reg} (r_reg = -1 for an outlier).  Returns a
// disabled plan for scenes the table does not serve (no sphere, more than 128, non-finite data).  `target_cells`: cells aimed at
// (cubic cells, 1..kDirMaskMaxDim per axis); `k`: bins per face edge.
inline DirMaskParams dirmask_plan(const float4 *sph, uint32_t n, uint32_t target_cells, uint32_t k, float4 *reg, bool take_outliers = true) {
  DirMaskParams dm{};
  if (n == 0 || n > kDirMaskMaxSpheres || k < 1 || k > kDirMaskMaxK || target_cells < 1) return dm;
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  double r[kDirMaskMaxSpheres];
  for (uint32_t i = 0; i < n; ++i) {
    const double c[3] = {sph[i].x, sph[i].y, sph[i].z};
    if (!(sph[i].w >= 0.0f) || !std::isfinite(sph[i].w) || !std::isfinite(c[0]) || !std::isfinite(c[1]) || !std::isfinite(c[2])) return dm;
  }
  bool outlier[kDirMaskMaxSpheres] = {};
  if (take_outliers) {
    dm.n_outliers = dirmask_outliers(sph, n, dm.outlier_slot);
    for (uint32_t j = 0; j < dm.n_outliers; ++j) {
      outlier[dm.outlier_slot[j]] = true;
      dm.outlier_mask[dm.outlier_slot[j] >> 6] |= 1ull << (dm.outlier_slot[j] & 63u);
    }
  }
  for (uint32_t i = 0; i < n; ++i) {
    const double c[3] = {sph[i].x, sph[i].y, sph[i].z};
    r[i] = std::sqrt(static_cast<double>(sph[i].w));
    if (outlier[i]) continue;
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
  double rr[kDirMaskMaxSpheres] = {}, glo[3], ghi[3], ext[3], edge = 0.0;
  uint32_t dims[3] = {1, 1, 1};
  for (int pass = 0; pass < 2; ++pass) {
    for (int a = 0; a < 3; ++a) {
      glo[a] = INFINITY;
      ghi[a] = -INFINITY;
    }
    for (uint32_t i = 0; i < n; ++i) {
      if (outlier[i]) continue;
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
  for (uint32_t i = 0; i < n; ++i) reg[i] = make_float4(sph[i].x, sph[i].y, sph[i].z, outlier[i] ? -1.0f : static_cast<float>(rr[i] * 1.0001));
  dm.enabled = 1;
  dm.table = nullptr;
  return dm;
}

// ---- Candidate lists: scenes of more than 128 spheres ---------------------------------------------------------------
// The same table [cell of the ray's origin][bin of its direction], but an entry is a LIST: 16 half-words,
//     h[0] = n << 12 | q      n <= 15 sphere slots follow in h[1 .. n] (ascending), q = the entry's reach in quanta
// and it is REACH-LIMITED: it holds every sphere the float test can accept at a distance of up to q * reach_unit from
// where the ray is looked up — nothing is promised beyond.  A ray is looked up at parameter s (0, or where it enters the box B),
// tests the entry's spheres — with its true origin and direction, so a sphere gives the same key in whichever round it
// is tested — and is done when its best hit lies within s + 0.97 * reach; otherwise it moves on by 0.95 * reach and looks
// the entry of the point it has reached up (or leaves B, and is done).  By induction every sphere accepted at a
// parameter up to s + reach has been tested after the round at s, so a best hit inside that range is the scan's result: any
// sphere not yet tested can only be accepted further out.  Dense neighbourhoods get short reaches (the builder cuts an
// entry's reach until 15 slots suffice), sparse ones long reaches; an entry that cannot be cut to 15 (more than 15
// registered balls overlap the cell) has q = 0: rays looked up there scan linearly, like every ray the table does not
// serve (direction not a unit vector, NaN, parameter beyond s_cap).
//
// Registered balls.  As for the masks, with two differences.  (1) The rounding and non-unit-direction term of the radius,
// 0.003 * D, is taken at D_i = 1.01 * (d_cap + r_i) with d_cap = s_cap + the largest reach, instead of at the near
// box's diagonal: lookups happen at s <= s_cap only, an accepted point lies within reach of the lookup point, so a sphere
// the float test accepts there has |o - c| <= (s_cap + reach) (1 + 2e-6) + r_reg <= D_i — and a sphere FARTHER from the
// origin than D_i can only be accepted beyond s + reach ((0.997 D - r) / (1 + 2e-6) > s_cap + reach), where nothing is
// promised.  In a scene of a thousand small spheres the near box's diagonal would add a third to every radius.  (2) Two
// exclusion tests instead of one, a sphere is listed unless either proves that no ray of the entry reaches its ball within
// the reach: the cone of rt_dirmask's masks (bin's angular radius + the cell's half diagonal), and the bin's PYRAMID
// swept over the cell: for a side plane n of the pyramid (n . d <= 0 for every direction of the grown bin) and any ray
// point x = o' + t d', n . (x - p) <= n . (o' - p) <= half edge * (|n_x| + |n_y| + |n_z|); likewise behind the cell along the
// bin's major axis.  The pyramid is what keeps the spheres beside and behind a cell out of its entries (a third fewer
// candidates on the 1 024-sphere scene, tools/large_scene_model.py).
constexpr uint32_t kDirListSlots = 15;
constexpr uint32_t kDirListMaxSpheres = 65535;  // slots are 16-bit
constexpr uint32_t kDirListMaxDim = 128;        // cells per axis
constexpr uint32_t kDirListReachQuanta = 4095;  // 12 bits
constexpr uint32_t kDirListWords = 4;           // 64-bit words per entry
constexpr float kDirListDone = 0.97f;           // a best hit within s + this * reach ends a ray's scan
constexpr float kDirListStep = 0.95f;           // otherwise the next lookup is this * reach further on

// What the builder needs of one (cell, bin): the cell's centre and extents (with slack), the bin's centre direction,
// cos / sin of its angular radius (grown), and the pyramid's four side planes.
struct DirListGeom {
  f3 p, a;
  float h, he, ct, st;
  f3 n[4];
  float support[4];
  uint32_t axis;
  float sign;
};

RT_HD DirListGeom dirlist_geometry(const DirMaskParams &dm, uint32_t cell, uint32_t bin) {
  DirListGeom g;
  const uint32_t ix = cell % dm.nx, iy = (cell / dm.nx) % dm.ny, iz = cell / (dm.nx * dm.ny);
  g.p = mk3(dm.bmin[0] + (static_cast<float>(ix) + 0.5f) * dm.cell[0], dm.bmin[1] + (static_cast<float>(iy) + 0.5f) * dm.cell[1],
            dm.bmin[2] + (static_cast<float>(iz) + 0.5f) * dm.cell[2]);
  const float edge = fmaxf(fmaxf(dm.cell[0], dm.cell[1]), dm.cell[2]);
  const float box = fmaxf(fmaxf(dm.bmax[0] - dm.bmin[0], dm.bmax[1] - dm.bmin[1]), dm.bmax[2] - dm.bmin[2]);
  // slack for the rounding of the lookup point o + s d (a few ulp of coordinates of the box's size: ~1e-7 box) and of its cell index
  const float slack = 1e-4f * box;
  g.he = 0.5f * edge * 1.01f + slack;
  g.h = g.he * 1.7320509f * 1.0001f;
  const uint32_t k = dm.k;
  const uint32_t bu = bin % k, bv = (bin / k) % k, face = bin / (k * k);
  g.axis = face >> 1;
  g.sign = (face & 1u) ? -1.0f : 1.0f;
  const float step = 2.0f / static_cast<float>(k), grow = 0.02f * step + 1e-4f;
  const float u0 = -1.0f + step * static_cast<float>(bu) - grow, u1 = -1.0f + step * static_cast<float>(bu + 1u) + grow;
  const float v0 = -1.0f + step * static_cast<float>(bv) - grow, v1 = -1.0f + step * static_cast<float>(bv + 1u) + grow;
  const uint32_t a0 = g.axis, a1 = (g.axis + 1u) % 3u, a2 = (g.axis + 2u) % 3u;
  const float sign = g.sign;
  auto vec = [&](float c0, float c1, float c2) {  // components along (major axis, u axis, v axis)
    float c[3];
    c[a0] = c0;
    c[a1] = c1;
    c[a2] = c2;
    return mk3(c[0], c[1], c[2]);
  };
  auto unit = [&](f3 v) {
    const float inv = 1.0f / sqrtf(v.x * v.x + v.y * v.y + v.z * v.z);
    return mk3(v.x * inv, v.y * inv, v.z * inv);
  };
  g.a = unit(vec(sign, 0.5f * (u0 + u1), 0.5f * (v0 + v1)));
  {
    const f3 c00 = unit(vec(sign, u0, v0)), c01 = unit(vec(sign, u0, v1)), c10 = unit(vec(sign, u1, v0)), c11 = unit(vec(sign, u1, v1));
    const f3 a = g.a;
    float cos_theta = fminf(fminf(a.x * c00.x + a.y * c00.y + a.z * c00.z, a.x * c01.x + a.y * c01.y + a.z * c01.z),
                            fminf(a.x * c10.x + a.y * c10.y + a.z * c10.z, a.x * c11.x + a.y * c11.y + a.z * c11.z));
    cos_theta = fminf(cos_theta, 1.0f);
    const float sin_theta0 = sqrtf(fmaxf(1.0f - cos_theta * cos_theta, 0.0f));
    g.ct = cos_theta * 0.999998f - sin_theta0 * 2e-3f;  // cos / sin of theta + 2e-3
    g.st = sin_theta0 * 0.999998f + cos_theta * 2e-3f;
  }
  // side planes: u = d_u / |d_major| <= u1  <=>  d_u - u1 sign d_major <= 0, and so on
  g.n[0] = unit(vec(-u1 * sign, 1.0f, 0.0f));
  g.n[1] = unit(vec(u0 * sign, -1.0f, 0.0f));
  g.n[2] = unit(vec(-v1 * sign, 0.0f, 1.0f));
  g.n[3] = unit(vec(v0 * sign, 0.0f, -1.0f));
  for (int j = 0; j < 4; ++j) g.support[j] = g.he * (fabsf(g.n[j].x) + fabsf(g.n[j].y) + fabsf(g.n[j].z)) * 1.0001f;
  return g;
}

// Can a ray of the entry be accepted by the sphere {c, r_reg} at all — and if so, `dmin`: a lower bound of the distance
// from the ray's lookup point to any point of the registered ball.  NaN anywhere -> yes, at distance 0.
RT_HD bool dirlist_candidate(const DirListGeom &g, float4 reg, float &dmin) {
  dmin = 0.0f;
  if (reg.w < 0.0f) return true;  // an outlier (registered with radius -1, "Outliers" above): in every list, at distance 0
  const float vx = reg.x - g.p.x, vy = reg.y - g.p.y, vz = reg.z - g.p.z;
  const float l2 = vx * vx + vy * vy + vz * vz;
  const float R = reg.w + g.h;
  dmin = 0.0f;
  if (!(l2 > R * R * 1.0001f)) return true;  // the cell (grown) touches the registered ball; NaN -> in
  const float l = sqrtf(l2);
  {  // the bin's cone
    const float inv_l = 1.0f / l;
    const float s = fminf(R * inv_l * 1.0001f, 1.0f);
    const float c = sqrtf(fmaxf(1.0f - s * s, 0.0f));
    const float cos_sum = g.ct * c - g.st * s;
    const float cos_ang = (g.a.x * vx + g.a.y * vy + g.a.z * vz) * inv_l;
    if (cos_ang < cos_sum - 1e-4f) return false;
  }
  // the bin's pyramid swept over the cell; the margin covers the rounding of the dot products (1e-6 of the box at most)
  const float margin = reg.w * 1.0001f + 1e-5f * l;
  for (int j = 0; j < 4; ++j)
    if (g.n[j].x * vx + g.n[j].y * vy + g.n[j].z * vz > g.support[j] + margin) return false;
  const float along = g.axis == 0u ? vx : (g.axis == 1u ? vy : vz);
  if (-g.sign * along > g.he + margin) return false;  // behind the cell along the major axis
  dmin = fmaxf(l * 0.9999f - R, 0.0f);
  return true;
}

// One list entry from the registered spheres recs[0 .. count) = {c, r_reg}; their slots are slot[j], or j when `slot` is
// null (ascending either way).  Host and device run this same function; the device hands in the spheres near the cell only
// (those with dmin <= the largest reach: the others cannot be listed).
RT_HD void dirlist_entry(const DirMaskParams &dm, uint32_t cell, uint32_t bin, const float4 *recs, const uint16_t *slot, uint32_t count,
                         uint32_t out[8]) {
  const DirListGeom g = dirlist_geometry(dm, cell, bin);
  const float reach_max = dm.reach_unit * static_cast<float>(kDirListReachQuanta);
  auto put = [&](uint32_t n, uint32_t j) {  // slot number n of the entry = sphere j of the list
    const uint32_t s = slot ? static_cast<uint32_t>(slot[j]) : j;
    const uint32_t hw = 1u + n;  // half-word index
    out[hw >> 1] |= s << (16u * (hw & 1u));
  };
  for (int w = 0; w < 8; ++w) out[w] = 0u;
  // One pass serves most entries: the candidates within the largest reach are written as they are found, up to 15.
  uint32_t n = 0;
  bool more = false;
  for (uint32_t j = 0; j < count; ++j) {
    float dmin;
    if (dirlist_candidate(g, recs[j], dmin) && dmin <= reach_max) {
      if (n < kDirListSlots) put(n++, j);
      else more = true;
    }
  }
  if (!more) {
    out[0] |= (n << 12) | kDirListReachQuanta;
    return;
  }
  // More than 15: the entry's reach is cut to the largest quantum count q with at most 15 candidates within q * reach_unit — i.e. with
  // q * reach_unit below the 16th smallest lower bound, found by a second pass that keeps the 16 smallest (q = 0: even the 16th
  // is 0 — more than 15 registered balls overlap the cell —: the entry serves no ray).
  float best[kDirListSlots + 1u];
  for (uint32_t k = 0; k <= kDirListSlots; ++k) best[k] = INFINITY;
  for (uint32_t j = 0; j < count; ++j) {
    float dmin;
    if (dirlist_candidate(g, recs[j], dmin) && dmin <= reach_max && dmin < best[kDirListSlots]) {
      uint32_t k = kDirListSlots;
      while (k > 0u && best[k - 1u] > dmin) {
        best[k] = best[k - 1u];
        k -= 1u;
      }
      best[k] = dmin;
    }
  }
  const float d16 = best[kDirListSlots];
  uint32_t q = 0u;
  if (d16 > 0.0f) {
    const float fq = floorf(d16 / dm.reach_unit);
    q = fq >= static_cast<float>(kDirListReachQuanta) ? kDirListReachQuanta : static_cast<uint32_t>(fq);
    while (q > 0u && !(dm.reach_unit * static_cast<float>(q) < d16)) q -= 1u;  // (the product as the kernels and the pass below form it)
  }
  for (int w = 0; w < 8; ++w) out[w] = 0u;
  if (q == 0u) return;  // n = 0, q = 0
  const float reach = dm.reach_unit * static_cast<float>(q);
  n = 0;
  for (uint32_t j = 0; j < count; ++j) {
    float dmin;
    if (dirlist_candidate(g, recs[j], dmin) && dmin <= reach && n < kDirListSlots) put(n++, j);
  }
  out[0] |= (n << 12) | q;
}

// Host side: resolution, boxes and registered radii for n spheres {cx, cy, cz, r*r}; reg[i] receives {c, r_reg}.  Returns a
// disabled plan for scenes the lists do not serve (no sphere, more than 65 535, non-finite data).
inline DirMaskParams dirlist_plan(const float4 *sph, uint32_t n, uint32_t target_cells, uint32_t k, uint32_t reach_cells, float4 *reg,
                                  bool take_outliers = true) {
  DirMaskParams dm{};
  if (n == 0 || n > kDirListMaxSpheres || k < 1 || k > kDirMaskMaxK || target_cells < 1 || reach_cells < 1) return dm;
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  std::vector<double> r(n);
  for (uint32_t i = 0; i < n; ++i)
    if (!(sph[i].w >= 0.0f) || !std::isfinite(sph[i].w) || !std::isfinite(sph[i].x) || !std::isfinite(sph[i].y) || !std::isfinite(sph[i].z)) return dm;
  // spheres far from all the others stay out of the lists, as out of the bit sets ("Outliers" above): boxes, d_cap and the rounding
  // terms are those of the rest, every ray tests them after its rounds (intersect_listed)
  std::vector<char> outlier(n, 0);
  if (take_outliers) {
    dm.n_outliers = dirmask_outliers(sph, n, dm.outlier_slot);
    for (uint32_t j = 0; j < dm.n_outliers; ++j) outlier[dm.outlier_slot[j]] = 1;
  }
  for (uint32_t i = 0; i < n; ++i) {
    const double c[3] = {sph[i].x, sph[i].y, sph[i].z};
    r[i] = std::sqrt(static_cast<double>(sph[i].w));
    if (outlier[i]) continue;
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::fmin(lo[a], c[a] - r[i]);
      hi[a] = std::fmax(hi[a], c[a] + r[i]);
    }
  }
  const double diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
  if (!(diag > 0.0) || !std::isfinite(diag) || diag > 1e15) return dm;
  std::vector<double> sorted(r);
  std::nth_element(sorted.begin(), sorted.begin() + n / 2, sorted.end());
  const double r_med = sorted[n / 2];
  // d_cap: how far from a ray's origin the table is asked about spheres (s_cap + the largest reach).  A hundred median
  // radii keep the rounding term of the registered radius (0.003 d_cap) below a third of a typical radius.
  const double d_cap = std::fmin(2.0 * diag, std::fmax(100.0 * r_med, 0.5 * diag));
  const double reach_bound = 0.3 * d_cap, s_cap = 0.7 * d_cap;
  std::vector<double> rr(n);
  double glo[3], ghi[3], ext[3], edge = 0.0;
  uint32_t dims[3] = {1, 1, 1};
  for (int pass = 0; pass < 2; ++pass) {
    for (int a = 0; a < 3; ++a) {
      glo[a] = INFINITY;
      ghi[a] = -INFINITY;
    }
    for (uint32_t i = 0; i < n; ++i) {
      if (outlier[i]) continue;
      const double eps = 0.003 * 1.01 * (d_cap + r[i]);
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
      dims[a] = static_cast<uint32_t>(std::fmin(static_cast<double>(kDirListMaxDim), std::fmax(1.0, std::ceil(want - 1e-9))));
    }
    edge = 0.0;
    for (int a = 0; a < 3; ++a) edge = std::fmax(edge, ext[a] / dims[a]);
  }
  for (int a = 0; a < 3; ++a) {
    dm.bmin[a] = static_cast<float>(glo[a]);
    dm.bmax[a] = static_cast<float>(ghi[a]);
    dm.cell[a] = static_cast<float>(edge);
    dm.inv_cell[a] = static_cast<float>(1.0 / edge);
    dm.near_min[a] = -INFINITY;  // (the lists bound a ray's distance by s_cap instead of a near box)
    dm.near_max[a] = INFINITY;
  }
  dm.nx = dims[0];
  dm.ny = dims[1];
  dm.nz = dims[2];
  dm.k = k;
  dm.bins = 6u * k * k;
  dm.words = kDirListWords;
  dm.entries = dm.nx * dm.ny * dm.nz * dm.bins;
  dm.half_k = 0.5f * static_cast<float>(k);
  dm.lists = 1;
  const double reach = std::fmin(static_cast<double>(reach_cells) * edge, reach_bound);
  dm.reach_unit = static_cast<float>(reach / static_cast<double>(kDirListReachQuanta) * (1.0 - 1e-6));
  dm.s_cap = static_cast<float>(s_cap);
  for (uint32_t i = 0; i < n; ++i) reg[i] = make_float4(sph[i].x, sph[i].y, sph[i].z, outlier[i] ? -1.0f : static_cast<float>(rr[i] * 1.0001));
  dm.enabled = 1;
  dm.table = nullptr;
  return dm;
}

}  // namespace rt355
