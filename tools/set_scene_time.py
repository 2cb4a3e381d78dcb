#!/usr/bin/env python3
"""What rt_set_scene costs with the direction-mask table (64 spheres) and with the candidate lists (1 024 spheres): first call
(table built on the device), a call with the same spheres (a moving camera: table kept), a call with one sphere moved (rebuilt).
python tools/set_scene_time.py [k] [cells]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, synth  # noqa: E402
from rsraytracingv3_amd import _native as N  # noqa: E402

k = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 0
dev = Device(0)
if k:
    dev.set_option(N.RT_OPT_DIRMASK_K, k)
if cells:
    dev.set_option(N.RT_OPT_DIRMASK_CELLS, cells)
W, H = 1920, 1080
scene = cases.headline_scene(8)
target = RenderTarget.new(dev, (W, H))
rt = RandomTexture(dev)
rt.write(synth.random_texture(1))
conn = Connection(scene, dev, synth.procedural_skybox(256, 128), rt, max_objects="live")


def timed(label):
    conn.last_size = (0, 0)  # force the dirty branch
    dev.sync()
    t0 = time.perf_counter()
    conn.update_buffers((W, H), scene, (0.0, 0.0))
    dev.sync()
    dt = time.perf_counter() - t0
    st = dev.stats()
    print("%-46s %8.3f ms   (table %d cells x %d bins)" % (label, dt * 1e3, st.dirmask_cells, st.dirmask_bins), flush=True)


def run_case(on):
    dev.set_option(N.RT_OPT_DIRMASK, on)
    n = len(scene.objects) - 1
    timed("%d spheres, tables %d: first rt_set_scene" % (n, on))
    conn.draw(); dev.sync()
    timed("%d spheres, tables %d: same spheres again" % (n, on))
    scene.camera.position = (56.0, 65.0, 55.0)
    timed("%d spheres, tables %d: camera moved" % (n, on))
    c = scene.objects[3].geometry.center
    scene.objects[3].geometry.center = (c[0] + 0.5, c[1], c[2])
    timed("%d spheres, tables %d: one sphere moved" % (n, on))
    conn.draw(); dev.sync()


for on in (1, 0):
    run_case(on)
scene = cases.stress_scene(1024, 8)  # candidate lists (tables 1) against the sphere grid (tables 0)
conn = Connection(scene, dev, synth.procedural_skybox(256, 128), rt, max_objects="live")
for on in (1, 0):
    run_case(on)
dev.close()
