// rt_grid.h — host-side build of the uniform sphere grid of large scenes (GridParams in rt_device.h).
//
// Registration radius.  The float sphere test (ray.hlsl:53-68 as specified in rt_math.h) accepts a sphere when
// its computed q = dn^2 - (|n|^2 - r^2) is >= 0 (and the root is in range); |q_float - q| <= 16 * 2^-24 * D^2 with
// D = |origin - centre| (three roundings in each dot product, one in each difference).  The shader's quadratic
// assumes a unit direction (utils.hlsl:5 is called with a = 1); for a direction with | |d|^2 - 1 | <= 4e-6 that adds
// at most 4e-6 * D^2 to the difference between q and the geometric r^2 - b^2, and the point the ray reaches at the
// accepted distance lies within sqrt(r^2 + 9e-6 * D^2) of the centre.  So with
//     r_reg = 1.001 * sqrt(r^2 + (0.003 * D_max)^2) + 0.01 * (smallest cell edge)
// every accepted (ray, sphere) pair has its accepted point inside the registered ball, for every ray that starts
// inside the "near" box (the spheres' bounds grown by a quarter of their diagonal; D_max = that box's diagonal)
// with such a direction.  Other rays — far origins, directions that are not unit vectors (a reflection off a plane
// whose stored normal is not normalised, ray.hlsl:43-45) — scan linearly.
// The last term covers the traversal's own rounding (reciprocal directions, accumulated boundary parameters).
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

#include "rt_device.h"

namespace rt355 {

struct GridBuild {
  GridParams params{};
  std::vector<uint32_t> cell_start;
  std::vector<uint16_t> items;
};

// Cells per sphere the resolution aims at.  A/B on one device (tools/grid_density_sweep.sh, Grays/s at 128 / 256 / 1024
// spheres): 2 cells per sphere 38.4 / 30.2 / 16.0, 1: 39.7 / 31.1 / 16.4, 0.5: 40.3 / 31.6 / 18.8, 0.35: 40.2 / 31.8 / 18.9,
// 0.125: 40.2 / 31.4 / 18.2, 0.06: 39.2 / 28.7 / 16.7 — a lane's walk costs per cell it crosses (and lanes of a wave cross
// different cells), a few more spheres per cell cost less.
constexpr double kGridCellsPerSphere = 0.35;
constexpr uint32_t kGridMaxDim = 64;
constexpr uint32_t kGridMaxItems = 12288;          // u16 entries (24 KB)
constexpr size_t kGridLdsBudget = 56u * 1024u;     // sphere records + cell_start + items

// sph: the padded sphere records {cx, cy, cz, r*r}; n_live of them are real.
inline GridBuild build_grid(const std::vector<float4> &sph, size_t n_live, size_t n_pad, uint32_t min_spheres) {
  GridBuild g;
  if (n_live < min_spheres || n_live > 65535) return g;
  std::vector<double> r(n_live);
  double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (size_t i = 0; i < n_live; ++i) {
    const double c[3] = {sph[i].x, sph[i].y, sph[i].z};
    if (!(sph[i].w >= 0.0f) || !std::isfinite(sph[i].w) || !std::isfinite(c[0]) || !std::isfinite(c[1]) || !std::isfinite(c[2]))
      return g;  // never-hit or non-finite records: scenes that odd scan linearly
    r[i] = std::sqrt(static_cast<double>(sph[i].w));
    for (int a = 0; a < 3; ++a) {
      lo[a] = std::min(lo[a], c[a] - r[i]);
      hi[a] = std::max(hi[a], c[a] + r[i]);
    }
  }
  const double diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
  if (!(diag > 0.0) || !std::isfinite(diag) || diag > 1e18) return g;
  double near_lo[3], near_hi[3], near_diag = 0.0;
  for (int a = 0; a < 3; ++a) {
    near_lo[a] = lo[a] - 0.25 * diag;
    near_hi[a] = hi[a] + 0.25 * diag;
    near_diag += (near_hi[a] - near_lo[a]) * (near_hi[a] - near_lo[a]);
  }
  near_diag = std::sqrt(near_diag);
  const double eps = 0.003 * near_diag;
  std::vector<double> rr(n_live);
  double glo[3] = {INFINITY, INFINITY, INFINITY}, ghi[3] = {-INFINITY, -INFINITY, -INFINITY};
  for (size_t i = 0; i < n_live; ++i) {
    rr[i] = 1.001 * std::sqrt(r[i] * r[i] + eps * eps);
    const double c[3] = {sph[i].x, sph[i].y, sph[i].z};
    for (int a = 0; a < 3; ++a) {
      glo[a] = std::min(glo[a], c[a] - rr[i]);
      ghi[a] = std::max(ghi[a], c[a] + rr[i]);
    }
  }
  double ext[3];
  for (int a = 0; a < 3; ++a) ext[a] = ghi[a] - glo[a];
  // resolution: kGridCellsPerSphere cells per sphere, cubic cells, at least one and at most kGridMaxDim per axis
  double target_cells = kGridCellsPerSphere * static_cast<double>(n_live);
  for (int attempt = 0; attempt < 6; ++attempt, target_cells *= 0.5) {
    const double s = std::cbrt(ext[0] * ext[1] * ext[2] / target_cells);
    uint32_t n[3];
    for (int a = 0; a < 3; ++a) {
      const double want = s > 0.0 ? ext[a] / s : 1.0;
      n[a] = static_cast<uint32_t>(std::min<double>(kGridMaxDim, std::max(1.0, std::floor(want + 0.5))));
    }
    double cell[3], min_cell = INFINITY;
    for (int a = 0; a < 3; ++a) {
      cell[a] = ext[a] / n[a];
      min_cell = std::min(min_cell, cell[a]);
    }
    const double slack = 0.01 * min_cell;
    // the box grows by the slack on every side; cells are recomputed over the grown box
    double blo[3], bhi[3];
    for (int a = 0; a < 3; ++a) {
      blo[a] = glo[a] - slack;
      bhi[a] = ghi[a] + slack;
      cell[a] = (bhi[a] - blo[a]) / n[a];
    }
    const uint32_t n_cells = n[0] * n[1] * n[2];
    std::vector<uint32_t> count(n_cells + 1, 0);
    std::vector<uint32_t> range(n_live * 6);
    uint64_t total = 0;
    for (size_t i = 0; i < n_live; ++i) {
      const double c[3] = {sph[i].x, sph[i].y, sph[i].z};
      uint64_t cells = 1;
      for (int a = 0; a < 3; ++a) {
        const double l = std::floor((c[a] - rr[i] - slack - blo[a]) / cell[a]), h = std::floor((c[a] + rr[i] + slack - blo[a]) / cell[a]);
        const uint32_t li = static_cast<uint32_t>(std::min<double>(n[a] - 1, std::max(0.0, l)));
        const uint32_t hj = static_cast<uint32_t>(std::min<double>(n[a] - 1, std::max(0.0, h)));
        range[i * 6 + a] = li;
        range[i * 6 + 3 + a] = hj;
        cells *= hj - li + 1;
      }
      total += cells;
    }
    const size_t lds = n_pad * 16 + (static_cast<size_t>(n_cells) + 1) * 4 + total * 2;
    if (total > kGridMaxItems || lds > kGridLdsBudget) continue;  // too many entries (huge spheres): coarser
    for (size_t i = 0; i < n_live; ++i)
      for (uint32_t z = range[i * 6 + 2]; z <= range[i * 6 + 5]; ++z)
        for (uint32_t y = range[i * 6 + 1]; y <= range[i * 6 + 4]; ++y)
          for (uint32_t x = range[i * 6]; x <= range[i * 6 + 3]; ++x) count[(z * n[1] + y) * n[0] + x + 1] += 1;
    for (uint32_t c = 0; c < n_cells; ++c) count[c + 1] += count[c];
    g.cell_start = count;
    g.items.assign(total, 0);
    std::vector<uint32_t> fill(count.begin(), count.end() - 1);
    for (size_t i = 0; i < n_live; ++i)  // ascending sphere slot inside every cell
      for (uint32_t z = range[i * 6 + 2]; z <= range[i * 6 + 5]; ++z)
        for (uint32_t y = range[i * 6 + 1]; y <= range[i * 6 + 4]; ++y)
          for (uint32_t x = range[i * 6]; x <= range[i * 6 + 3]; ++x) g.items[fill[(z * n[1] + y) * n[0] + x]++] = static_cast<uint16_t>(i);
    GridParams &p = g.params;
    p.enabled = 1;
    p.nx = n[0];
    p.ny = n[1];
    p.nz = n[2];
    p.n_cells = n_cells;
    p.n_items = static_cast<uint32_t>(total);
    for (int a = 0; a < 3; ++a) {
      p.gmin[a] = static_cast<float>(blo[a]);
      p.gmax[a] = static_cast<float>(bhi[a]);
      p.cell[a] = static_cast<float>(cell[a]);
      p.inv_cell[a] = static_cast<float>(1.0 / cell[a]);
      // the near box shrinks by a rounding's worth, so that "inside" in float implies inside in double
      p.near_min[a] = static_cast<float>(near_lo[a] + 1e-6 * near_diag);
      p.near_max[a] = static_cast<float>(near_hi[a] - 1e-6 * near_diag);
    }
    p.t_margin = static_cast<float>(1e-4 * near_diag);
    return g;
  }
  return g;  // disabled
}

}  // namespace rt355
