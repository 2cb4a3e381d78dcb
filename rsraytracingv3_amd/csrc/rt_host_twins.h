// rt_host_twins.h — the host-side twins of the structures rt_set_scene builds (sphere grid, direction-mask table, candidate
// lists), behind the C-ABI of include/rt355.h: plain host code on rt_grid.h / rt_dirmask.h / rt_math.h, for inspection and for
// testing the structures without a GPU.  Included by rt_shim.hip (inside its extern "C" block) and by host_asan.cpp, which builds
// them — with the EXR reader — under -fsanitize=address,undefined (`make -C rsraytracingv3_amd/csrc host-asan`).
// No #pragma once: it defines functions; include it exactly once per library.

int rt_grid_build_host(const float *spheres_xyzr2, uint32_t n_spheres, uint32_t min_spheres, rt_grid_info *info, uint32_t *cell_start,
                       uint32_t cell_cap, uint16_t *items, uint32_t item_cap) {
  if (!info || (!spheres_xyzr2 && n_spheres)) return RT_ERR_INVALID_ARGUMENT;
  try {
    const size_t n_pad = (static_cast<size_t>(n_spheres) + kSphBatch - 1) / kSphBatch * kSphBatch + kSphBatch;
    std::vector<float4> sph(n_pad, make_float4(0.0f, 0.0f, 0.0f, -INFINITY));
    for (uint32_t i = 0; i < n_spheres; ++i)
      sph[i] = make_float4(spheres_xyzr2[4 * i], spheres_xyzr2[4 * i + 1], spheres_xyzr2[4 * i + 2], spheres_xyzr2[4 * i + 3]);
    const GridBuild g = build_grid(sph, n_spheres, n_pad, min_spheres);
    std::memset(info, 0, sizeof *info);
    info->enabled = g.params.enabled;
    if (!g.params.enabled) return RT_OK;
    info->nx = g.params.nx;
    info->ny = g.params.ny;
    info->nz = g.params.nz;
    info->n_cells = g.params.n_cells;
    info->n_items = g.params.n_items;
    for (int a = 0; a < 3; ++a) {
      info->gmin[a] = g.params.gmin[a];
      info->gmax[a] = g.params.gmax[a];
      info->cell[a] = g.params.cell[a];
      info->inv_cell[a] = g.params.inv_cell[a];
      info->near_min[a] = g.params.near_min[a];
      info->near_max[a] = g.params.near_max[a];
    }
    info->t_margin = g.params.t_margin;
    if (cell_start && cell_cap >= g.cell_start.size()) std::memcpy(cell_start, g.cell_start.data(), g.cell_start.size() * 4);
    if (items && item_cap >= g.items.size()) std::memcpy(items, g.items.data(), g.items.size() * 2);
    return RT_OK;
  } catch (...) {
    return RT_ERR_OUT_OF_MEMORY;
  }
}

// Host-side twin of the direction-mask table (rt_dirmask.h), for inspection and for testing its sets without a GPU.
static DirMaskParams dirmask_from_info(const rt_dirmask_info *info) {
  DirMaskParams dm{};
  dm.enabled = info->enabled;
  dm.nx = info->nx;
  dm.ny = info->ny;
  dm.nz = info->nz;
  dm.k = info->k;
  dm.bins = info->bins;
  dm.words = info->words;
  dm.entries = info->nx * info->ny * info->nz * info->bins;
  dm.half_k = 0.5f * static_cast<float>(info->k);
  for (int a = 0; a < 3; ++a) {
    dm.bmin[a] = info->bmin[a];
    dm.bmax[a] = info->bmax[a];
    dm.cell[a] = info->cell[a];
    dm.inv_cell[a] = info->inv_cell[a];
    dm.near_min[a] = info->near_min[a];
    dm.near_max[a] = info->near_max[a];
  }
  dm.n_outliers = info->n_outliers < kDirMaskMaxOutliers ? info->n_outliers : kDirMaskMaxOutliers;
  for (uint32_t j = 0; j < dm.n_outliers; ++j) dm.outlier_slot[j] = info->outlier_slot[j];
  return dm;
}

int rt_dirmask_build_host(const float *spheres_xyzr2, uint32_t n_spheres, uint32_t target_cells, uint32_t k, rt_dirmask_info *info,
                          uint64_t *table, size_t table_cap) {
  if (!info || (!spheres_xyzr2 && n_spheres)) return RT_ERR_INVALID_ARGUMENT;
  std::memset(info, 0, sizeof *info);
  if (n_spheres == 0 || n_spheres > kDirMaskMaxSpheres) return RT_OK;  // not served: info->enabled stays 0
  float4 sph[kDirMaskMaxSpheres], reg[kDirMaskMaxSpheres];
  for (uint32_t i = 0; i < n_spheres; ++i)
    sph[i] = make_float4(spheres_xyzr2[4 * i], spheres_xyzr2[4 * i + 1], spheres_xyzr2[4 * i + 2], spheres_xyzr2[4 * i + 3]);
  const DirMaskParams dm = dirmask_plan(sph, n_spheres, target_cells, k, reg);
  if (!dm.enabled) return RT_OK;
  info->enabled = 1;
  info->nx = dm.nx;
  info->ny = dm.ny;
  info->nz = dm.nz;
  info->k = dm.k;
  info->bins = dm.bins;
  info->words = dm.words;
  for (int a = 0; a < 3; ++a) {
    info->bmin[a] = dm.bmin[a];
    info->bmax[a] = dm.bmax[a];
    info->cell[a] = dm.cell[a];
    info->inv_cell[a] = dm.inv_cell[a];
    info->near_min[a] = dm.near_min[a];
    info->near_max[a] = dm.near_max[a];
  }
  info->n_outliers = dm.n_outliers;
  for (uint32_t j = 0; j < dm.n_outliers; ++j) info->outlier_slot[j] = dm.outlier_slot[j];
  const size_t entries = static_cast<size_t>(dm.nx) * dm.ny * dm.nz * dm.bins;
  if (table && table_cap >= entries * dm.words)
    for (size_t e = 0; e < entries; ++e) {
      unsigned long long w[2];
      dirmask_entry(dm, static_cast<uint32_t>(e / dm.bins), static_cast<uint32_t>(e % dm.bins), reg, n_spheres, w);
      table[e * dm.words] = w[0];
      if (dm.words > 1u) table[e * dm.words + 1u] = w[1];
    }
  return RT_OK;
}

int rt_dirmask_lookup_host(const rt_dirmask_info *info, const uint64_t *table, const float *origin3, const float *direction3,
                           uint64_t *mask2, int *served) {
  if (!info || !table || !origin3 || !direction3 || !mask2 || !served || !info->enabled) return RT_ERR_INVALID_ARGUMENT;
  const DirMaskParams dm = dirmask_from_info(info);
  const f3 o = mk3(origin3[0], origin3[1], origin3[2]), d = mk3(direction3[0], direction3[1], direction3[2]);
  mask2[0] = mask2[1] = 0;
  *served = 0;
  // the lane's arithmetic (dirmask_lookup in rt_kernels.hip), with an IEEE reciprocal where the device uses v_rcp_f32
  const bool ok = (fabsf(dot3(d, d) - 1.0f) <= 4e-6f) && (o.x >= dm.near_min[0]) && (o.x <= dm.near_max[0]) && (o.y >= dm.near_min[1]) &&
                  (o.y <= dm.near_max[1]) && (o.z >= dm.near_min[2]) && (o.z <= dm.near_max[2]);
  if (!ok) return RT_OK;
  *served = 1;
  auto add_outliers = [&] {
    for (uint32_t j = 0; j < dm.n_outliers; ++j) mask2[dm.outlier_slot[j] >> 6] |= 1ull << (dm.outlier_slot[j] & 63u);
  };
  f3 p = o;
  const bool outside = !((o.x >= dm.bmin[0]) && (o.x <= dm.bmax[0]) && (o.y >= dm.bmin[1]) && (o.y <= dm.bmax[1]) && (o.z >= dm.bmin[2]) &&
                         (o.z <= dm.bmax[2]));
  if (outside) {
    const float big = 1e30f;
    const float ivx = fabsf(d.x) > 1e-30f ? 1.0f / d.x : (d.x < 0.0f ? -big : big);
    const float ivy = fabsf(d.y) > 1e-30f ? 1.0f / d.y : (d.y < 0.0f ? -big : big);
    const float ivz = fabsf(d.z) > 1e-30f ? 1.0f / d.z : (d.z < 0.0f ? -big : big);
    const float ax = (dm.bmin[0] - o.x) * ivx, bx = (dm.bmax[0] - o.x) * ivx;
    const float ay = (dm.bmin[1] - o.y) * ivy, by = (dm.bmax[1] - o.y) * ivy;
    const float az = (dm.bmin[2] - o.z) * ivz, bz = (dm.bmax[2] - o.z) * ivz;
    const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    if (!(t_in <= t_out)) {  // misses the box: the device reads the extra entry that holds the outliers alone
      add_outliers();
      return RT_OK;
    }
    p = fma3(d, t_in, o);
  }
  const uint32_t cell = dirmask_cell(dm, p);
  const float inv_major = 1.0f / fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
  const size_t entry = static_cast<size_t>(cell) * dm.bins + dirmask_bin(d, inv_major, dm.k, dm.half_k);
  mask2[0] = table[entry * dm.words];  // (every entry holds the outliers)
  if (dm.words > 1u) mask2[1] = table[entry * dm.words + 1u];
  return RT_OK;
}

// Host-side twin of the candidate lists (rt_dirmask.h "Candidate lists"), for testing them without a GPU: the plan, single
// entries on demand (a whole table is hundreds of megabytes), and a lane's walk through its rounds.
static DirMaskParams dirlist_from_info(const rt_dirlist_info *info) {
  DirMaskParams dm{};
  dm.enabled = info->enabled;
  dm.nx = info->nx;
  dm.ny = info->ny;
  dm.nz = info->nz;
  dm.k = info->k;
  dm.bins = info->bins;
  dm.words = kDirListWords;
  dm.entries = info->nx * info->ny * info->nz * info->bins;
  dm.half_k = 0.5f * static_cast<float>(info->k);
  dm.lists = 1;
  dm.reach_unit = info->reach_unit;
  dm.s_cap = info->s_cap;
  for (int a = 0; a < 3; ++a) {
    dm.bmin[a] = info->bmin[a];
    dm.bmax[a] = info->bmax[a];
    dm.cell[a] = info->cell[a];
    dm.inv_cell[a] = info->inv_cell[a];
  }
  dm.n_outliers = info->n_outliers < kDirMaskMaxOutliers ? info->n_outliers : kDirMaskMaxOutliers;
  for (uint32_t j = 0; j < dm.n_outliers; ++j) dm.outlier_slot[j] = info->outlier_slot[j];
  return dm;
}

int rt_dirlist_plan_host(const float *spheres_xyzr2, uint32_t n_spheres, uint32_t target_cells, uint32_t k, uint32_t reach_cells,
                         rt_dirlist_info *info, float *registered_xyzr) {
  if (!info || (!spheres_xyzr2 && n_spheres)) return RT_ERR_INVALID_ARGUMENT;
  std::memset(info, 0, sizeof *info);
  try {
    std::vector<float4> sph(n_spheres), reg(n_spheres);
    for (uint32_t i = 0; i < n_spheres; ++i)
      sph[i] = make_float4(spheres_xyzr2[4 * i], spheres_xyzr2[4 * i + 1], spheres_xyzr2[4 * i + 2], spheres_xyzr2[4 * i + 3]);
    const DirMaskParams dm = dirlist_plan(sph.data(), n_spheres, target_cells, k, reach_cells, reg.data());
    if (!dm.enabled) return RT_OK;
    info->enabled = 1;
    info->nx = dm.nx;
    info->ny = dm.ny;
    info->nz = dm.nz;
    info->k = dm.k;
    info->bins = dm.bins;
    info->reach_unit = dm.reach_unit;
    info->s_cap = dm.s_cap;
    for (int a = 0; a < 3; ++a) {
      info->bmin[a] = dm.bmin[a];
      info->bmax[a] = dm.bmax[a];
      info->cell[a] = dm.cell[a];
      info->inv_cell[a] = dm.inv_cell[a];
    }
    info->n_outliers = dm.n_outliers;
    for (uint32_t j = 0; j < dm.n_outliers; ++j) info->outlier_slot[j] = dm.outlier_slot[j];
    if (registered_xyzr) std::memcpy(registered_xyzr, reg.data(), static_cast<size_t>(n_spheres) * sizeof(float4));
    return RT_OK;
  } catch (...) {
    return RT_ERR_OUT_OF_MEMORY;
  }
}

int rt_dirlist_entry_host(const rt_dirlist_info *info, const float *registered_xyzr, uint32_t n_spheres, uint32_t cell, uint32_t bin,
                          uint32_t *entry8) {
  if (!info || !info->enabled || !registered_xyzr || !entry8 || cell >= info->nx * info->ny * info->nz || bin >= info->bins)
    return RT_ERR_INVALID_ARGUMENT;
  const DirMaskParams dm = dirlist_from_info(info);
  dirlist_entry(dm, cell, bin, reinterpret_cast<const float4 *>(registered_xyzr), nullptr, n_spheres, entry8);
  return RT_OK;
}

// One ray through the rounds of intersect_listed (rt_kernels.hip), with IEEE reciprocals where the device uses v_rcp_f32 and
// entries computed on demand: *listed_key / *listed_slot = what the rounds find (slot -1: nothing; *far = 1: the ray is not
// served and the kernel scans linearly — then they are the linear result), *linear_key / *linear_slot = the linear scan of
// all spheres with the same arithmetic; `t_plane`: the distance of the ray's plane hit (a large value for none), which ends the
// rounds like a sphere hit does.  The two results must be equal for every ray.
int rt_dirlist_walk_host(const rt_dirlist_info *info, const float *spheres_xyzr2, const float *registered_xyzr, uint32_t n_spheres,
                         const float *origin3, const float *direction3, float t_plane, uint32_t *listed_key, int32_t *listed_slot,
                         uint32_t *linear_key, int32_t *linear_slot, uint32_t *rounds, uint32_t *tests, int *far) {
  if (!info || !info->enabled || !spheres_xyzr2 || !registered_xyzr || !origin3 || !direction3 || !listed_key || !listed_slot ||
      !linear_key || !linear_slot || !rounds || !tests || !far)
    return RT_ERR_INVALID_ARGUMENT;
  constexpr uint32_t kEps = 0x358637bdu, kNone = 0x49742400u - kEps;  // as in rt_kernels.hip
  const DirMaskParams dm = dirlist_from_info(info);
  const f3 o = mk3(origin3[0], origin3[1], origin3[2]), d = mk3(direction3[0], direction3[1], direction3[2]);
  auto test = [&](uint32_t i, uint32_t &key, int32_t &slot) {  // one sphere of ray.hlsl:53-68, the kernels' arithmetic
    const float *g = spheres_xyzr2 + 4 * static_cast<size_t>(i);
    const float nx = o.x - g[0], ny = o.y - g[1], nz = o.z - g[2];
    const float dn = fmaf(d.z, nz, fmaf(d.y, ny, d.x * nx));
    const float q = fmaf(dn, dn, -(fmaf(nz, nz, fmaf(ny, ny, nx * nx)) - g[3]));
    if (fminf(q, -dn) >= 0.0f) {
      const uint32_t k = f2u(-dn - sqrtf(q)) - kEps;
      if (k < key || (k == key && k < kNone && static_cast<int32_t>(i) < slot)) {
        if (k < kNone) {
          key = k;
          slot = static_cast<int32_t>(i);
        }
      }
    }
  };
  *linear_key = kNone;
  *linear_slot = -1;
  for (uint32_t i = 0; i < n_spheres; ++i) test(i, *linear_key, *linear_slot);
  *listed_key = kNone;
  *listed_slot = -1;
  *rounds = 0;
  *tests = 0;
  *far = 0;
  const bool unit = fabsf(dot3(d, d) - 1.0f) <= 4e-6f;
  const bool inside = (o.x >= dm.bmin[0]) && (o.x <= dm.bmax[0]) && (o.y >= dm.bmin[1]) && (o.y <= dm.bmax[1]) && (o.z >= dm.bmin[2]) &&
                      (o.z <= dm.bmax[2]);
  bool todo = unit, is_far = !unit;
  float s = 0.0f;
  if (todo && !inside) {
    const float big = 1e30f;
    const float ivx = fabsf(d.x) > 1e-30f ? 1.0f / d.x : (d.x < 0.0f ? -big : big);
    const float ivy = fabsf(d.y) > 1e-30f ? 1.0f / d.y : (d.y < 0.0f ? -big : big);
    const float ivz = fabsf(d.z) > 1e-30f ? 1.0f / d.z : (d.z < 0.0f ? -big : big);
    const float ax = (dm.bmin[0] - o.x) * ivx, bx = (dm.bmax[0] - o.x) * ivx;
    const float ay = (dm.bmin[1] - o.y) * ivy, by = (dm.bmax[1] - o.y) * ivy;
    const float az = (dm.bmin[2] - o.z) * ivz, bz = (dm.bmax[2] - o.z) * ivz;
    const float t_in = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fmaxf(fminf(az, bz), 0.0f));
    const float t_out = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
    const bool finite = (o.x == o.x) && (o.y == o.y) && (o.z == o.z);
    const bool misses = finite && !(t_in <= t_out);
    const bool beyond = !finite || (t_in > dm.s_cap);
    // (with outliers — listed by every entry — a ray that misses the box still looks the entry up where it starts: intersect_listed)
    const bool keep_missing = dm.n_outliers != 0u;
    s = (misses && keep_missing) ? 0.0f : t_in;
    is_far = beyond;
    todo = !(misses && !keep_missing) && !beyond;
  }
  const float inv_major = 1.0f / fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fabsf(d.z));
  const uint32_t bin = unit ? dirmask_bin(d, inv_major, dm.k, dm.half_k) : 0u;
  while (todo) {
    if (*rounds == 48u) {
      is_far = true;
      break;
    }
    *rounds += 1;
    const f3 p = fma3(d, s, o);
    uint32_t e[8];
    dirlist_entry(dm, dirmask_cell(dm, p), bin, reinterpret_cast<const float4 *>(registered_xyzr), nullptr, n_spheres, e);
    const uint32_t n = (e[0] >> 12) & 15u, q = e[0] & 0xfffu;
    for (uint32_t j = 0; j < n; ++j) {
      const uint32_t hw = 1u + j;
      test((e[hw >> 1] >> (16u * (hw & 1u))) & 0xffffu, *listed_key, *listed_slot);
      *tests += 1;
    }
    const float t_sphere = u2f(*listed_key + kEps);
    const float t_near = fminf(t_plane, t_sphere);
    const float reach = static_cast<float>(q) * dm.reach_unit;
    if (q == 0u) {
      is_far = true;
      todo = false;
    } else if (t_near <= fmaf(reach, kDirListDone, s)) {
      todo = false;
    } else {
      s = fmaf(reach, kDirListStep, s);
      const f3 pn = fma3(d, s, o);
      const bool in_box = (pn.x >= dm.bmin[0]) && (pn.x <= dm.bmax[0]) && (pn.y >= dm.bmin[1]) && (pn.y <= dm.bmax[1]) &&
                          (pn.z >= dm.bmin[2]) && (pn.z <= dm.bmax[2]);
      if (!in_box) todo = false;
      else if (s > dm.s_cap) {
        is_far = true;
        todo = false;
      }
    }
  }
  if (is_far) {
    *far = 1;
    *listed_key = *linear_key;
    *listed_slot = *linear_slot;
  }
  return RT_OK;
}

