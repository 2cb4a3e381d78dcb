#!/usr/bin/env python3
"""Differential fuzzing of the HIP path against the CPU oracle (single frames, and every second case a
multi-frame launch against the oracle frame chain): random scenes (overlapping spheres,
tiny and huge radii, spheres around the camera, several planes, mirror / rough / emissive materials),
random cameras, sizes, bounce limits, jitters, kernel shapes; every frame must be bit-identical and the
in-kernel culling self-check must stay at zero.   python tools/fuzz_parity.py [cases] [first_seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle  # noqa: E402
from rsraytracingv3_amd import Connection, Device, MultiDevice, RandomTexture, RenderTarget, render_scene, synth  # noqa: E402
from rsraytracingv3_amd import _native as N  # noqa: E402
from rsraytracingv3_amd.scene import Camera, Geometry, Material, Object, Scene  # noqa: E402


def random_case(seed):
    rs = np.random.RandomState(seed)
    n_sph = int(rs.choice([0, 1, 3, 17, 64, 65, 100, 128, 130, 300, 700]))
    extent = float(rs.choice([5.0, 60.0, 400.0]))
    objs = []
    for i in range(n_sph):
        r = float(np.exp(rs.uniform(np.log(0.05), np.log(extent / (3 if n_sph < 200 or seed % 3 == 0 else 12)))))
        c = tuple(float(v) for v in rs.uniform(-extent, extent, 3))
        rough = float(rs.choice([0.0, 0.0, rs.uniform(0, 1), 1.0]))
        emis = float(rs.choice([0.0, 0.0, rs.uniform(0, 12)]))
        objs.append(Object("s%d" % i, Material(tuple(map(float, rs.uniform(0, 1, 3))), tuple(map(float, rs.randint(0, 2, 3))),
                                              emis, 0.5, rough), Geometry.sphere(c, r)))
    if n_sph >= 12 and rs.randint(0, 4) == 0:  # spheres far from all the others: taken out of the candidate tables (rt_dirmask.h "Outliers")
        for j in range(int(rs.randint(1, 7))):  # up to four are taken, the others stay in
            c = tuple(float(v) for v in rs.uniform(-1.0, 1.0, 3) * extent * rs.choice([6.0, 15.0, 40.0]))
            objs.insert(int(rs.randint(0, len(objs) + 1)),
                        Object("far%d" % j, Material(tuple(map(float, rs.uniform(0, 1, 3))), (1.0, 1.0, 1.0), float(rs.choice([0.0, 2.0])), 0.5,
                                                     float(rs.choice([0.0, 0.4]))), Geometry.sphere(c, float(extent * rs.choice([0.01, 0.5, 3.0])))))
    for j in range(int(rs.choice([0, 1, 1, 3]))):
        nrm = rs.normal(size=3)
        nrm = tuple(float(v) for v in (nrm / np.linalg.norm(nrm) * rs.choice([1.0, 1.0, 2.5])))
        objs.insert(int(rs.randint(0, len(objs) + 1)),
                    Object("p%d" % j, Material(tuple(map(float, rs.uniform(0, 1, 3))), (1.0, 1.0, 1.0), float(rs.choice([0.0, 0.3])),
                                               0.2, float(rs.choice([0.0, 0.5]))),
                           Geometry.plane(tuple(float(v) for v in rs.uniform(-extent / 2, extent / 2, 3)), nrm,
                                          float(rs.choice([extent / 4, 1e5])))))
    cam = Camera(tuple(float(v) for v in rs.uniform(-extent, extent, 3)),
                 (float(rs.uniform(-1.5, 1.5)), float(rs.uniform(-3.1, 3.1)), float(rs.uniform(-0.5, 0.5))),
                 float(rs.choice([1.0, 20.0, 70.0, 90.0])))
    scene = Scene(objects=objs, camera=cam, reflection_limit=int(rs.choice([1, 2, 3, 8, 20])))
    size = (int(rs.randint(1, 200)), int(rs.randint(1, 120)))
    if rs.randint(0, 6) == 0:  # a grid of 8 or 16 workgroups across: multi-frame launches of render_blocks use the XCD band map
        size = (int(rs.choice([121, 128, 249, 256])), int(rs.randint(1, 300)))
    jitter = (float(rs.uniform(-0.495, 0.495)), float(rs.uniform(-0.495, 0.495)))
    slots = "live" if (len(objs) > 80 or rs.rand() < 0.5) else None
    return scene, size, jitter, slots, int(rs.choice([-1, 0, 1, 2, 3])), int(rs.randint(1, 1000))


def grid_min_for(seed):
    """RT_OPT_GRID_MIN of a case: every third case forces the sphere grid on small scenes too."""
    return 1 if seed % 3 == 1 else 129


def dirmask_for(seed):
    """RT_OPT_DIRMASK* of a case: the direction-mask table at every resolution, for tiny scenes too; every fifth case off."""
    return {N.RT_OPT_DIRMASK: 0 if seed % 5 == 4 else 1, N.RT_OPT_DIRMASK_MIN: 1,
            N.RT_OPT_DIRMASK_K: (1, 2, 4, 8, 16)[(seed // 3) % 5], N.RT_OPT_DIRMASK_CELLS: (1, 27, 512, 4096)[(seed // 7) % 4]}


def dirlist_for(seed):
    """RT_OPT_DIRLIST* of a case: candidate lists at every resolution and reach, every fourth case for small scenes too (from
    one sphere up); off where the case forces the sphere grid (grid_min_for) and in every eleventh case."""
    on = 0 if (seed % 3 == 1 or seed % 11 == 5) else 1
    return {N.RT_OPT_DIRLIST: on, N.RT_OPT_DIRLIST_MIN: 1 if seed % 4 == 2 else 129,
            N.RT_OPT_DIRLIST_K: (1, 2, 4, 8, 12)[(seed // 5) % 5], N.RT_OPT_DIRLIST_CELLS: (1, 27, 512, 16384, 100000)[(seed // 3) % 5],
            N.RT_OPT_DIRLIST_REACH: (1, 2, 8, 32, 128)[(seed // 2) % 5]}


def run(cases=100, first=0):
    dev = Device(0)
    multi = MultiDevice([0, 0, 0], 8)  # every seventh case also goes through three contexts of one process (rt_multi_*)
    sky = synth.procedural_skybox(128, 64)
    bad = grids = tables = multis = spills = lists = 0
    for seed in range(first, first + cases):
        scene, size, jitter, slots, variant, tex_seed = random_case(seed)
        rnd = synth.random_texture(tex_seed)
        dev.set_partition(0, 1, 8)
        dev.set_accum_mode(N.RT_ACCUM_SRGB8)
        dev.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32 | N.RT_OUT_SEGMENTS | N.RT_OUT_HITS)
        dev.set_variant(variant)
        dev.set_option(N.RT_OPT_CULL, 2)
        dev.set_option(N.RT_OPT_GRID_MIN, grid_min_for(seed))
        for opt, val in list(dirmask_for(seed).items()) + list(dirlist_for(seed).items()):
            dev.set_option(opt, val)
        target = RenderTarget.new(dev, size)
        rt = RandomTexture(dev)
        rt.write(rnd)
        conn = Connection(scene, dev, sky, rt, max_objects=slots)
        render_scene(dev, conn, target, scene, jitter)
        dev.sync()
        st = dev.stats()
        w, h = size
        if slots == "live":
            objects, count = scene.live_object_bytes(), len(scene.objects)
        else:
            objects, count = scene.as_bytes(w, h)[0], 80
        ref = oracle.render(objects, count, scene.config_bytes(w, h), jitter, 0, sky, rnd, threads=8)
        ok = (np.array_equal(target.read_rgbf32().view(np.uint32), ref.rgbf32.view(np.uint32))
              and np.array_equal(target.read_rgba8(), ref.rgba8)
              and np.array_equal(target.read_segments(), ref.segments)
              and np.array_equal(target.read_hits(scene.reflection_limit), ref.hits)
              and st.last_cull_mismatches == 0 and st.last_segments == int(ref.segments.sum()))
        if ok and seed % 2 == 0:  # several frames in one launch against the oracle's frame chain
            rs = np.random.RandomState(seed + 77)
            nf, pc0, accum = int(rs.randint(2, 7)), int(rs.choice([0, 0, 5])), int(rs.choice([0, 1]))
            jit = rs.uniform(-0.495, 0.495, (nf, 2)).astype(np.float32)
            dev.set_accum_mode(accum)
            dev.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32)
            if seed % 4 == 0:
                dev.set_variant(3)  # every second multi-frame launch takes the re-packing shape whatever the case drew
            if variant == 3 or seed % 4 == 0:  # sample-only launches of the re-packing shape: the tail spill at several thresholds
                dev.set_option(N.RT_OPT_CULL, 1)  # (the self-check frames of RT_OPT_CULL = 2 are not sample-only)
                dev.set_option(N.RT_OPT_FLAGS, 4)  # launches of every size spill
                dev.set_option(N.RT_OPT_TAIL_SPILL, (4, 32, 128)[(seed // 2) % 3])
                dev.set_option(N.RT_OPT_REFILL_MIN, (1, 16, 64)[(seed // 6) % 3])
            target = RenderTarget.new(dev, size)
            conn.draw_frames(jit, pc0)
            dev.sync()
            spills += dev.stats().last_stage >= 100
            dev.set_option(N.RT_OPT_CULL, 2)
            dev.set_option(N.RT_OPT_FLAGS, 0)
            dev.set_option(N.RT_OPT_TAIL_SPILL, N.RT_TAIL_SPILL_DEFAULT)
            dev.set_option(N.RT_OPT_REFILL_MIN, 1)
            hist8 = np.zeros((h, w, 4), np.uint8) if accum == 0 else None
            histf = np.zeros((h, w, 3), np.float32) if accum == 1 else None
            for k in range(nf):
                ref = oracle.render(objects, count, scene.config_bytes(w, h), (float(jit[k, 0]), float(jit[k, 1])), pc0 + k,
                                    sky, rnd, history_rgba8=hist8, history_f32=histf, accum_mode=accum, threads=8,
                                    want_hits=False)
                hist8, histf = (ref.rgba8, None) if accum == 0 else (None, ref.rgbf32)
            ok = (np.array_equal(target.read_rgba8(), ref.rgba8)
                  and np.array_equal(target.read_rgbf32().view(np.uint32), ref.rgbf32.view(np.uint32)))
            if not ok:
                print("  (in the %d-frame launch, first count %d, accum %d)" % (nf, pc0, accum), flush=True)
        if ok and seed % 7 == 3:  # the same frame, its row strips dealt over three contexts and gathered on the first
            multis += 1
            for opt, val in list(dirmask_for(seed).items()) + list(dirlist_for(seed).items()):
                multi.set_option(opt, val)
            multi.set_option(N.RT_OPT_GRID_MIN, grid_min_for(seed))
            multi.set_accum_mode(N.RT_ACCUM_SRGB8)
            RenderTarget.new(multi, size)
            rtm = RandomTexture(multi)
            rtm.write(rnd)
            connm = Connection(scene, multi, sky, rtm, max_objects=slots)
            connm.update_buffers(size, scene, jitter)
            connm.frame_data.jitter, connm.frame_data.progressive_count = jitter, 0
            connm.draw()
            first_ref = oracle.render(objects, count, scene.config_bytes(w, h), jitter, 0, sky, rnd, threads=8, want_hits=False)
            ok = np.array_equal(multi.read_frame_rgba8(w, h), first_ref.rgba8)
            if not ok:
                print("  (through the three-context rt_multi)", flush=True)
        if not ok:
            bad += 1
            print("MISMATCH seed %d: %d objects, size %s, R %d, variant %d, slots %s, cull mismatches %d, grid cells %d, mask table %d x %d"
                  % (seed, len(scene.objects), size, scene.reflection_limit, variant, slots, st.last_cull_mismatches, st.grid_cells,
                     st.dirmask_cells, st.dirmask_bins),
                  flush=True)
        grids += st.last_stage == 3
        tables += st.dirmask_cells != 0 and st.dirmask_lists == 0
        lists += st.dirmask_lists != 0
        if (seed - first + 1) % 1000 == 0:  # a long soak must not look hung to the box's watchdog
            print("  ... %d cases, %d mismatching so far" % (seed - first + 1, bad), flush=True)
    dev.set_option(N.RT_OPT_CULL, 1)
    dev.set_option(N.RT_OPT_GRID_MIN, 129)
    dev.set_option(N.RT_OPT_DIRLIST_MIN, 129)
    dev.close()
    multi.close()
    print("fuzz: %d cases from seed %d (%d of them walked a sphere grid, %d used a direction-mask table, %d candidate lists, %d multi-frame "
          "launches spilled their tail, %d also went through a three-context rt_multi), %d mismatching"
          % (cases, first, grids, tables, lists, spills, multis, bad), flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    f = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    sys.exit(1 if run(n, f) else 0)
