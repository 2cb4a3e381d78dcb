"""The direction-mask table (rsraytracingv3_amd/csrc/rt_dirmask.h) on the CPU tier.

The kernels test a secondary ray only against the spheres of its table entry, so an entry must hold EVERY sphere the
linear scan could accept for a ray that maps to it.  Checked here with the host twin of the table (the same entry
function the device kernel runs, ``rt_dirmask_build_host`` / ``rt_dirmask_lookup_host``) against the oracle's own
rays: for every executed iteration of every pixel (``oracle.render_with_rays``), the sphere the oracle's linear scan
hit must be a member of the ray's set.  (The GPU tier repeats this inside the kernels with RT_OPT_CULL = 2.)"""
import ctypes as C

import numpy as np
import pytest

import cases
import oracle
from rsraytracingv3_amd import _native as N
from rsraytracingv3_amd.scene import Camera


def build(spheres, target_cells, k):
    lib = N.load()
    sph = np.ascontiguousarray(spheres, np.float32)
    info = N.RtDirMaskInfo()
    assert lib.rt_dirmask_build_host(sph.ctypes.data_as(C.c_void_p), len(sph), target_cells, k, C.byref(info), None, 0) == 0
    if not info.enabled:
        return info, None
    table = np.zeros(info.nx * info.ny * info.nz * info.bins * info.words, np.uint64)
    assert lib.rt_dirmask_build_host(sph.ctypes.data_as(C.c_void_p), len(sph), target_cells, k, C.byref(info),
                                     table.ctypes.data_as(C.c_void_p), table.size) == 0
    return info, table


def lookup(info, table, o, d):
    lib = N.load()
    o = np.ascontiguousarray(o, np.float32)
    d = np.ascontiguousarray(d, np.float32)
    mask, served = (C.c_uint64 * 2)(0, 0), C.c_int(0)
    assert lib.rt_dirmask_lookup_host(C.byref(info), table.ctypes.data_as(C.c_void_p), o.ctypes.data_as(C.c_void_p),
                                      d.ctypes.data_as(C.c_void_p), mask, C.byref(served)) == 0
    return int(mask[0]) | (int(mask[1]) << 64), bool(served.value)


def sphere_records(scene):
    out, slots = [], {}
    for i, o in enumerate(scene.objects):
        if o.geometry.kind == 0:
            slots[i] = len(out)
            out.append((*o.geometry.center, np.float32(o.geometry.radius) * np.float32(o.geometry.radius)))
    return np.array(out, np.float32), slots


def check_scene(scene, size, target_cells, k, jitter=(0.1, -0.2)):
    w, h = size
    sky, rnd = cases.inputs(64, 32)
    frame, rays = oracle.render_with_rays(scene.live_object_bytes(), len(scene.objects), scene.config_bytes(w, h), jitter, 0,
                                          sky, rnd, threads=4)
    sph, slots = sphere_records(scene)
    info, table = build(sph, target_cells, k)
    assert info.enabled
    checked = served_n = members = 0
    for y in range(h):
        for x in range(w):
            for i in range(1, int(frame.segments[y, x])):  # secondary iterations only: primary rays use the tile masks
                o, d = rays[y, x, i, :3], rays[y, x, i, 3:]
                mask, served = lookup(info, table, o, d)
                checked += 1
                if not served:
                    continue
                served_n += 1
                members += bin(mask).count("1")
                obj = int(frame.hits[y, x, i])
                if obj in slots:
                    assert (mask >> slots[obj]) & 1, "pixel (%d, %d) iteration %d: sphere slot %d hit but not in the set %016x" % (
                        x, y, i, slots[obj], mask)
    return checked, served_n, members / max(served_n, 1)


@pytest.mark.parametrize("target_cells,k", [(64, 2), (512, 4), (4096, 8)])
def test_every_hit_sphere_is_in_its_rays_set_headline_scene(target_cells, k):
    scene = cases.headline_scene(8)
    checked, served, mean = check_scene(scene, (96, 54), target_cells, k)
    assert checked > 4000 and served > 0.9 * checked
    if k == 8:
        assert mean < 8.0  # and the sets are small: that is the point (64 spheres in the scene)


def test_sets_hold_for_other_views_and_scenes():
    # looking along the ground from inside the disc: long shallow rays, origins outside the spheres' box
    scene = cases.headline_scene(6)
    scene.camera = Camera((70.0, 4.0, 10.0), (0.02, -1.6, 0.0), 80.0)
    check_scene(scene, (64, 36), 1024, 6)
    # from far above: origins far from the scene
    scene.camera = Camera((5.0, 400.0, 5.0), (1.5, 0.0, 0.0), 40.0)
    check_scene(scene, (48, 27), 256, 4)
    # a small scene with larger spheres
    small = cases.spheres_scene(16, 8, seed=7, min_r=2.0, max_r=12.0, placement=30.0, request=30)
    check_scene(small, (64, 36), 2048, 8)


def test_spheres_far_from_all_the_others_stay_out_of_the_table():
    """rt_dirmask.h "Outliers": up to four spheres far from the rest do not size the table — the cells and the registered radii are those of
    the remaining spheres — and are members of EVERY set instead, also of the one rays that miss the box read.  The headline
    scene with three far spheres, two of them in view and one a mirror the paths bounce off: every sphere the oracle hit is in its
    ray's set, the table's box is the cluster's, and the sets stay as small as without the outliers."""
    from rsraytracingv3_amd.scene import Geometry, Material, Object
    base = cases.headline_scene(6)
    _, _, mean_base = check_scene(base, (64, 36), 4096, 8)
    info_base, _ = build(sphere_records(base)[0], 4096, 8)
    scene = cases.headline_scene(6)
    mirror = Material((0.9, 0.9, 0.9), (0, 0, 0), 0.0, 0.5, 0.0)
    far = [Geometry.sphere((-420.0, 60.0, -380.0), 90.0), Geometry.sphere((-900.0, 500.0, 100.0), 1.0), Geometry.sphere((30.0, 900.0, 30.0), 25.0)]
    for k, g in enumerate(far):
        scene.objects.insert(10 * k + 3, Object("far%d" % k, mirror, g))
    sph, slots = sphere_records(scene)
    info, table = build(sph, 4096, 8)
    assert info.enabled and info.n_outliers == 3 and [info.outlier_slot[j] for j in range(3)] == [3, 13, 23]
    for a in range(3):  # the cluster's box, not the bounding box of everything
        assert abs(info.bmin[a] - info_base.bmin[a]) < 1.0 and abs(info.bmax[a] - info_base.bmax[a]) < 1.0
    checked, served, mean = check_scene(scene, (64, 36), 4096, 8)
    assert served > 0.85 * checked and mean < mean_base + 3.0 + 0.5
    # a ray that misses the box gets the outliers alone
    mask, ok = lookup(info, table, (0.0, float(info.bmax[1]) + 5.0, 0.0), (0.0, 1.0, 0.0))
    assert ok and mask == (1 << 3) | (1 << 13) | (1 << 23)
    # scenes without outliers: none taken
    assert info_base.n_outliers == 0
    assert build(sphere_records(cases.stress_scene(100, 4))[0], 4096, 8)[0].n_outliers == 0


def test_the_reference_80_slot_scene_uses_two_word_sets():
    """connection.rs keeps 80 object slots, the unused ones zero-filled: radius-0 spheres at the origin the shader still
    tests.  64 spheres + 15 of those = 79 spheres -> two 64-bit words per entry."""
    scene = cases.headline_scene(8)
    w, h = 80, 45
    sky, rnd = cases.inputs(64, 32)
    objects = scene.as_bytes(w, h)[0]
    frame, rays = oracle.render_with_rays(objects, 80, scene.config_bytes(w, h), (0.0, 0.0), 0, sky, rnd, threads=4)
    import struct
    recs, slots = [], {}
    for i in range(80):
        cx, cy, cz = struct.unpack_from("<fff", objects, 112 * i + 64)   # include/rt355_layout.h: geometry.center @64
        option = struct.unpack_from("<I", objects, 112 * i + 48)[0]      # geometry.option @48
        f32 = struct.unpack_from("<f", objects, 112 * i + 92)[0]         # geometry.f32_data @92
        if option == 0:
            slots[i] = len(recs)
            recs.append((cx, cy, cz, np.float32(f32) * np.float32(f32)))
    assert len(recs) == 79
    info, table = build(np.array(recs, np.float32), 1024, 6)
    assert info.enabled and info.words == 2
    for y in range(h):
        for x in range(w):
            for i in range(1, int(frame.segments[y, x])):
                mask, served = lookup(info, table, rays[y, x, i, :3], rays[y, x, i, 3:])
                obj = int(frame.hits[y, x, i])
                if served and obj in slots:
                    assert (mask >> slots[obj]) & 1


def test_rays_the_table_does_not_serve_are_flagged():
    sph, _ = sphere_records(cases.headline_scene(2))
    info, table = build(sph, 256, 4)
    far = np.array([1e5, 3.0, 0.0], np.float32)
    assert lookup(info, table, far, [-1.0, 0.0, 0.0])[1] is False             # origin outside the near box
    assert lookup(info, table, [0.0, 30.0, 0.0], [0.0, -1.1, 0.0])[1] is False  # direction not a unit vector
    assert lookup(info, table, [np.nan, 0.0, 0.0], [0.0, -1.0, 0.0])[1] is False
    mask, served = lookup(info, table, [0.0, 40.0, 0.0], [0.0, 1.0, 0.0])        # inside the near box, heading away
    assert served and mask == 0
    mask, served = lookup(info, table, [0.0, 40.0, 0.0], [0.0, -1.0, 0.0])       # straight down onto the disc
    assert served and mask != 0


def test_scenes_that_are_not_served():
    info, table = build(np.zeros((0, 4), np.float32), 256, 4)
    assert not info.enabled and table is None
    too_many = np.tile(np.array([[0, 0, 0, 1]], np.float32), (129, 1))
    assert not build(too_many, 256, 4)[0].enabled
    bad = np.array([[0, 0, 0, 1], [np.inf, 0, 0, 1]], np.float32)
    assert not build(bad, 256, 4)[0].enabled
