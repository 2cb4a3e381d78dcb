#!/usr/bin/env python3
"""What the 65th sphere costs: the headline scene (64 spheres: one-word candidate sets) against the same scene with ONE more sphere
that no ray can see (far behind the camera) or a zero-radius one at the origin (the reference's zero-filled tail after rt_set_scene
has dropped the repeats), i.e. two-word sets for the same rays.  8 frames per launch, two launches in flight, wall time."""
import copy
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, synth  # noqa: E402
from rsraytracingv3_amd import _native as N  # noqa: E402
from rsraytracingv3_amd.scene import Geometry, Material, Object  # noqa: E402

dev = Device(0)
dev.set_outputs(N.RT_OUT_RGBA8)
rt = RandomTexture(dev)
rt.write(synth.random_texture(1))
sky = synth.procedural_skybox(2048, 1024)
jit = synth.jitter_sequence(42, 4096)
size = (1920, 1080)
RenderTarget.new(dev, size)


def measure(scene, max_objects, frames=640, batch=8):
    dev.set_option(N.RT_OPT_FLAGS, 32 if max_objects == "live+32" else 0)
    max_objects = "live" if max_objects == "live+32" else max_objects
    conn = Connection(scene, dev, sky, rt, max_objects=max_objects)
    conn.update_buffers(size, scene, (0.0, 0.0))
    best = 1e9
    for rep in range(4):
        dev.sync()
        t0 = time.perf_counter()
        for k in range(0, frames, batch):
            conn.draw_frames(jit[k:k + batch], k)
        dev.sync()
        best = min(best, (time.perf_counter() - t0) / frames)
    st = dev.stats()
    return st.last_segments / st.last_batch / best / 1e9, st


base = cases.headline_scene(8)
mat = Material((0.5, 0.5, 0.5), (0, 0, 0), 0.0, 0.5, 0.5)


def plus(geoms):
    s = copy.deepcopy(base)
    for i, g in enumerate(geoms):
        s.objects.insert(len(s.objects) - 1, Object("extra%d" % i, mat, g))
    return s


def minus_one(extra):
    s = copy.deepcopy(base)
    del s.objects[63]  # the 64th sphere
    for i, g in enumerate(extra):
        s.objects.insert(len(s.objects) - 1, Object("extra%d" % i, mat, g))
    return s


rows = [("64 spheres (live count)", base, "live"),
        ("63 spheres", minus_one([]), "live"),
        ("63 spheres + 1 of radius 0 at the origin (one-word sets)", minus_one([Geometry.sphere((0.0, 0.0, 0.0), 0.0)]), "live"),
        ("+ 1 sphere of radius 0 at the origin", plus([Geometry.sphere((0.0, 0.0, 0.0), 0.0)]), "live"),
        ("+ 1 sphere of radius 1 at (-500, 500, -500), behind the camera (an outlier: tested by every ray)", plus([Geometry.sphere((-500.0, 500.0, -500.0), 1.0)]), "live"),
        ("the same, the outlier kept in the table (RT_OPT_FLAGS 32)", plus([Geometry.sphere((-500.0, 500.0, -500.0), 1.0)]), "live+32"),
        ("+ 3 far spheres", plus([Geometry.sphere((-500.0, 500.0, -500.0), 1.0), Geometry.sphere((-420.0, 60.0, -380.0), 90.0), Geometry.sphere((30.0, 900.0, 30.0), 25.0)]), "live"),
        ("the reference's 80 slots (zero-filled tail: 15 -> 1 record)", base, None)]
for rep in range(2):
    for name, scene, mo in rows:
        g, st = measure(scene, mo)
        print("%-72s %6.2f Grays/s   (%d spheres scanned, table %d cells x %d bins)" % (name, g, st.spheres_scanned, st.dirmask_cells, st.dirmask_bins), flush=True)
dev.close()
