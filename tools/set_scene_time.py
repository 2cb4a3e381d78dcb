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
    """Python's serialisation of the scene (scene.rs:236-273 as_bytes — the host language's business), then the C call alone:
    until it returns, and until the device is idle again (the table build and the uploads are asynchronous)."""
    t0 = time.perf_counter()
    objects, count = conn._object_blob(scene, W, H)
    lights = scene.as_bytes(W, H)[1]
    config = scene.config_bytes(W, H)
    t_ser = time.perf_counter() - t0
    dev.sync()
    t0 = time.perf_counter()
    dev.call("rt_set_scene", objects, count, lights, type(scene).MAX_LIGHTS, config)
    t_ret = time.perf_counter() - t0
    dev.sync()
    t_idle = time.perf_counter() - t0
    st = dev.stats()
    print("%-46s rt_set_scene returns after %7.3f ms, device idle after %7.3f ms   (table %d cells x %d bins; serialising the scene "
          "in Python before it: %.3f ms)" % (label, t_ret * 1e3, t_idle * 1e3, st.dirmask_cells, st.dirmask_bins, t_ser * 1e3), flush=True)


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
