#!/usr/bin/env python3
"""Candidate structures for scenes of more than 128 spheres, priced on the oracle's rays before any kernel was written
(CPU only).  For every secondary ray of a config-3 frame (1 024 spheres) it counts the sphere tests of

  * a direction table [cell of the origin][cube-map bin of the direction] -> LIST of spheres (the structure of
    rt_dirmask.h without its 128-sphere limit), optionally with a REACH: an entry only lists the spheres a ray can be
    accepted by within `reach` of the cell, and a ray whose best hit lies beyond looks its next set up further along;
  * the uniform grid of rt_grid.h (cells crossed, spheres tested) — what render_tiles<3> walks today.

    python tools/large_scene_model.py [n_spheres] [width] [height]
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import cases  # noqa: E402
import oracle  # noqa: E402

n_sph = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
w = int(sys.argv[2]) if len(sys.argv) > 2 else 240
h = int(sys.argv[3]) if len(sys.argv) > 3 else 135
scene = cases.stress_scene(n_sph, 8) if n_sph > 128 else cases.spheres_scene(n_sph, 8, request=max(100, 2 * n_sph))
sky, rnd = cases.inputs(64, 32)
frame, rays = oracle.render_with_rays(scene.live_object_bytes(), len(scene.objects), scene.config_bytes(w, h), (0.1, -0.2), 0, sky, rnd,
                                      threads=8)
seg = frame.segments.astype(np.int64)
print("spheres %d  %dx%d  segments/pixel %.3f" % (n_sph, w, h, seg.mean()))
sph = np.array([(*o.geometry.center, o.geometry.radius) for o in scene.objects if o.geometry.kind == 0], np.float64)
C, Rr = sph[:, :3], sph[:, 3]
slot_of = {}
k_ = 0
for i, o in enumerate(scene.objects):
    if o.geometry.kind == 0:
        slot_of[i] = k_
        k_ += 1
# secondary rays
sel = []
for i in range(1, 8):
    m = seg > i
    ys, xs = np.nonzero(m)
    for y, x in zip(ys, xs):
        sel.append((y, x, i))
sel = np.array(sel)
O = rays[sel[:, 0], sel[:, 1], sel[:, 2], :3].astype(np.float64)
D = rays[sel[:, 0], sel[:, 1], sel[:, 2], 3:].astype(np.float64)
HIT = frame.hits[sel[:, 0], sel[:, 1], sel[:, 2]]
n_rays = len(O)
hit_sphere = np.array([h_ in slot_of for h_ in HIT])
print("secondary rays %d: hit a sphere %.3f, the plane %.3f, nothing %.3f" % (
    n_rays, hit_sphere.mean(), np.mean([(h_ >= 0) and (h_ not in slot_of) for h_ in HIT]), np.mean(HIT < 0)))
# true hit distance for sphere hits (geometric)
lo, hi = (C - Rr[:, None]).min(0), (C + Rr[:, None]).max(0)
diag = np.linalg.norm(hi - lo)
near_lo, near_hi = lo - 0.25 * diag, hi + 0.25 * diag
near_diag = np.linalg.norm(near_hi - near_lo)
eps = float(os.environ.get("MODEL_EPS", 0.003)) * near_diag
SLACK = float(os.environ.get("MODEL_SLACK", 2e-3))
FRUSTUM = int(os.environ.get("MODEL_FRUSTUM", 0))
print("sphere box", lo, hi, "diag %.1f  eps %.2f" % (diag, eps))


def plan(target_cells):
    edge = 0.0
    for _ in range(2):
        rr = 1.001 * np.sqrt(Rr * Rr + eps * eps) + 1e-3 * edge
        glo, ghi = (C - rr[:, None]).min(0), (C + rr[:, None]).max(0)
        bd = np.linalg.norm(ghi - glo)
        glo, ghi = glo - 1e-3 * bd, ghi + 1e-3 * bd
        ext = ghi - glo
        edge = (ext.prod() / target_cells) ** (1 / 3)
        dims = np.minimum(MAXDIM, np.maximum(1, np.ceil(ext / edge - 1e-9))).astype(int)
        edge = (ext / dims).max()
    return glo, ghi, dims, edge, rr * 1.0001


def bin_of(d, k):
    a = np.abs(d)
    ax = np.where((a[:, 0] >= a[:, 1]) & (a[:, 0] >= a[:, 2]), 0, np.where(a[:, 1] >= a[:, 2], 1, 2))
    idx = np.arange(len(d))
    major = d[idx, ax]
    u = d[idx, (ax + 1) % 3] / np.abs(major)
    v = d[idx, (ax + 2) % 3] / np.abs(major)
    face = 2 * ax + (major < 0)
    bu = np.minimum(k - 1, np.maximum(0, ((u + 1) * k / 2).astype(int)))
    bv = np.minimum(k - 1, np.maximum(0, ((v + 1) * k / 2).astype(int)))
    return (face * k + bv) * k + bu


def entry_set(p, hdiag, box, bin_, k, rr, reach):
    """dirmask_entry of rt_dirmask.h for all spheres at once, + an optional reach."""
    bu, bv, face = bin_ % k, (bin_ // k) % k, bin_ // (k * k)
    axis, sign = face >> 1, (-1.0 if face & 1 else 1.0)
    step = 2.0 / k
    grow = 0.02 * step + 1e-4
    u0, u1 = -1 + step * bu - grow, -1 + step * (bu + 1) + grow
    v0, v1 = -1 + step * bv - grow, -1 + step * (bv + 1) + grow

    def dirv(u, v):
        c = np.zeros(3)
        c[axis] = sign
        c[(axis + 1) % 3] = u
        c[(axis + 2) % 3] = v
        return c / np.linalg.norm(c)
    a = dirv(0.5 * (u0 + u1), 0.5 * (v0 + v1))
    cos_t = min(1.0, min(a @ dirv(u0, v0), a @ dirv(u0, v1), a @ dirv(u1, v0), a @ dirv(u1, v1)))
    sin_t = math.sqrt(max(0.0, 1 - cos_t * cos_t))
    ct, st = cos_t * 0.999998 - sin_t * 2e-3, sin_t * 0.999998 + cos_t * 2e-3
    hh = 0.5 * hdiag * 1.01 + SLACK * box
    v = C - p
    l2 = (v * v).sum(1)
    R = rr + hh
    inside = ~(l2 > R * R * 1.0001)
    l = np.sqrt(np.maximum(l2, 1e-30))
    s = np.minimum(R / l * 1.0001, 1.0)
    c = np.sqrt(np.maximum(1 - s * s, 0))
    cos_sum = ct * c - st * s
    cos_ang = (v @ a) / l
    m = inside | ~(cos_ang < cos_sum - 1e-4)
    if FRUSTUM:
        # the rays of the entry stay inside the bin's pyramid swept over the cell (a cube of half edge he): five planes
        he = hdiag / math.sqrt(3) / 2 * 1.01 + SLACK * box
        def nrm(axis_coef, other, other_coef):
            n = np.zeros(3)
            n[axis] = axis_coef
            n[other] = other_coef
            return n / np.linalg.norm(n)
        a1, a2 = (axis + 1) % 3, (axis + 2) % 3
        planes = [nrm(-u1 * sign, a1, 1.0), nrm(u0 * sign, a1, -1.0), nrm(-v1 * sign, a2, 1.0), nrm(v0 * sign, a2, -1.0)]
        back = np.zeros(3)
        back[axis] = -sign
        planes.append(back)
        for n in planes:
            m &= ~((v @ n) > np.abs(n).sum() * he + rr + 1e-4)
    if reach is not None:
        m &= l <= reach + R  # the accepted point lies within `reach` of the origin, the origin within hh of p
    return np.nonzero(m)[0]


MAXDIM = 64


def table_model(target_cells, k, reach_cells=None, maxdim=64, verbose=True):
    global MAXDIM
    MAXDIM = maxdim
    glo, ghi, dims, edge, rr = plan(target_cells)
    box = (ghi - glo).max()
    hdiag = math.sqrt(3) * edge
    reach = None if reach_cells is None else reach_cells * edge
    inside_near = np.all((O >= near_lo) & (O <= near_hi), 1)
    cache = {}
    tests = np.zeros(n_rays, np.int64)
    lookups = np.zeros(n_rays, np.int64)
    missing = 0
    binsD = bin_of(D, k)
    for r in range(n_rays):
        if not inside_near[r]:
            tests[r] = len(C)
            continue
        o, d = O[r], D[r]
        # entry into the box
        inv = 1.0 / np.where(np.abs(d) > 1e-30, d, 1e-30)
        t0, t1 = (glo - o) * inv, (ghi - o) * inv
        t_in = max(np.minimum(t0, t1).max(), 0.0)
        t_out = np.maximum(t0, t1).min()
        if not (t_in <= t_out):
            continue
        s = t_in
        best_t = math.inf
        if hit_sphere[r]:
            sl = slot_of[int(HIT[r])]
            b = (C[sl] - o) @ d
            best_t = b - math.sqrt(max(0.0, b * b - ((C[sl] - o) @ (C[sl] - o) - Rr[sl] ** 2)))
        found = False
        while True:
            p = o + s * d
            cell = np.minimum(dims - 1, np.maximum(0, ((p - glo) / edge).astype(int)))
            key = (cell[0], cell[1], cell[2], binsD[r])
            if key not in cache:
                centre = glo + (cell + 0.5) * edge
                cache[key] = entry_set(centre, hdiag, box, binsD[r], k, rr, reach)
            cand = cache[key]
            tests[r] += len(cand)
            lookups[r] += 1
            if hit_sphere[r] and sl in cand:
                found = True
            if reach is None:
                break
            if best_t <= s + 0.9 * reach:
                break
            s += 0.9 * reach
            if s > t_out:
                break
        if hit_sphere[r] and not found:
            missing += 1
    n_entries = int(dims.prod()) * 6 * k * k
    lens = np.array([len(v) for v in cache.values()])
    if verbose:
        print("table cells %s edge %.2f k %d reach %s: entries %.2fM; tests/ray mean %.2f  p50 %d p90 %d p99 %d max %d; lookups/ray %.3f; "
              "entry length mean %.1f max %d (over %d used entries); hit sphere not listed: %d; rays outside near box %d" % (
                  dims, edge, k, reach_cells, n_entries / 1e6, tests.mean(), np.percentile(tests, 50), np.percentile(tests, 90),
                  np.percentile(tests, 99), tests.max(), lookups.mean(), lens.mean(), lens.max(), len(lens), missing, (~inside_near).sum()))
    n = n_rays // 64 * 64
    print("   lookups per wave of 64 rays: max over lanes %.2f (all-done rounds); used entries longer than 7: %.3f" % (
        lookups[:n].reshape(-1, 64).max(1).mean(), (lens > 7).mean()))
    return tests


def wave_cost(tests, label):
    """what a wave pays: rays in groups of 64 in array order (neighbouring pixels of one bounce): balanced = ceil(sum / 64) trips,
    per-lane = the heaviest lane"""
    n = len(tests) // 64 * 64
    t = tests[:n].reshape(-1, 64)
    print("   %s: per wave of 64 rays: balanced trips %.2f, heaviest lane %.1f" % (label, np.ceil(t.sum(1) / 64).mean(), t.max(1).mean()))


def grid_model():
    import ctypes as Ct
    from rsraytracingv3_amd import _native as N
    lib = N.load()
    rec = np.ascontiguousarray(np.concatenate([C, (Rr * Rr)[:, None]], 1), np.float32)
    info = N.RtGridInfo()
    assert lib.rt_grid_build_host(rec.ctypes.data_as(Ct.c_void_p), len(rec), 1, Ct.byref(info), None, 0, None, 0) == 0
    if not info.enabled:
        print("grid: not enabled")
        return
    cs = np.zeros(info.n_cells + 1, np.uint32)
    it = np.zeros(info.n_items, np.uint16)
    lib.rt_grid_build_host(rec.ctypes.data_as(Ct.c_void_p), len(rec), 1, Ct.byref(info), cs.ctypes.data_as(Ct.c_void_p), len(cs),
                           it.ctypes.data_as(Ct.c_void_p), len(it))
    gmin, gmax = np.array(info.gmin[:]), np.array(info.gmax[:])
    cell = np.array(info.cell[:])
    dims = np.array([info.nx, info.ny, info.nz])
    tests = np.zeros(n_rays, np.int64)
    steps = np.zeros(n_rays, np.int64)
    for r in range(n_rays):
        o, d = O[r], D[r]
        inv = 1.0 / np.where(np.abs(d) > 1e-30, d, 1e-30)
        t0, t1 = (gmin - o) * inv, (gmax - o) * inv
        t_in = max(np.minimum(t0, t1).max(), 0.0)
        t_out = np.maximum(t0, t1).min()
        if not (t_in <= t_out):
            continue
        best_t = math.inf
        if hit_sphere[r]:
            sl = slot_of[int(HIT[r])]
            b = (C[sl] - o) @ d
            best_t = b - math.sqrt(max(0.0, b * b - ((C[sl] - o) @ (C[sl] - o) - Rr[sl] ** 2)))
        p = o + t_in * d
        ic = np.minimum(dims - 1, np.maximum(0, np.floor((p - gmin) / cell).astype(int)))
        sgn = np.where(d >= 0, 1, -1)
        tnext = np.maximum(((ic + (sgn > 0)) * cell + gmin - o) * inv, t_in)
        dt = cell * np.abs(inv)
        seen_best = math.inf
        while True:
            c = (ic[2] * dims[1] + ic[1]) * dims[0] + ic[0]
            items = it[cs[c]:cs[c + 1]]
            tests[r] += len(items)
            steps[r] += 1
            if hit_sphere[r] and sl in items:
                seen_best = best_t
            t_exit = tnext.min()
            if seen_best < t_exit - info.t_margin:
                break
            a = int(np.argmin(tnext))
            ic[a] += sgn[a]
            tnext[a] += dt[a]
            if ic[a] < 0 or ic[a] >= dims[a]:
                break
    print("grid %s cell %s: tests/ray mean %.2f p50 %d p90 %d p99 %d max %d; cells/ray mean %.2f p90 %d max %d" % (
        dims, np.round(cell, 2), tests.mean(), np.percentile(tests, 50), np.percentile(tests, 90), np.percentile(tests, 99), tests.max(),
        steps.mean(), np.percentile(steps, 90), steps.max()))
    wave_cost(tests, "grid tests")
    wave_cost(steps, "grid cells")


if __name__ == "__main__":
    grid_model()
    configs = [(4096, 12, 4, 64), (4096, 8, 8, 64), (8192, 8, 6, 128), (16384, 8, 4, 128), (16384, 8, 8, 128), (16384, 8, 16, 128),
               (16384, 6, 8, 128), (32768, 8, 8, 128)]
    if os.environ.get("MODEL_CONFIGS"):
        configs = eval(os.environ["MODEL_CONFIGS"])
    for cells, k, reach, md in configs:
        t = table_model(cells, k, reach, md)
        wave_cost(t, "table")
