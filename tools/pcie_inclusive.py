#!/usr/bin/env python3
"""PCIe-inclusive frame rate at the C-ABI boundary: render one 1080p frame and bring its RGBA8 texels to host memory, every
frame — what the reference's loop does when it shows each frame (render_target.rs:50-70, app.rs:240-253):

  * render only (no read-back): the rate to compare with;
  * rt_present_async / rt_present_wait: frame k's copy into a pinned staging slot runs on a copy stream beside frame k + 1's
    rendering (two slots alternate);
  * rt_read_rgba8 into pageable host memory after every frame (blocking), with and without a fresh 600x600 random texture
    uploaded per frame (random_texture.rs:56-75 does so every 4 ms).

    python tools/pcie_inclusive.py [frames]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cases  # noqa: E402
from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, synth  # noqa: E402
from rsraytracingv3_amd import _native as N  # noqa: E402

W, H = 1920, 1080
dev = Device(0)
dev.set_outputs(N.RT_OUT_RGBA8)
target = RenderTarget.new(dev, (W, H))
rt = RandomTexture(dev)
textures = [synth.random_texture(s) for s in range(4)]
rt.write(textures[0])
scene = cases.headline_scene(8)
conn = Connection(scene, dev, synth.procedural_skybox(2048, 1024), rt, max_objects="live")
conn.update_buffers((W, H), scene, (0.0, 0.0))
FRAMES = int(sys.argv[1]) if len(sys.argv) > 1 else 400
jit = synth.jitter_sequence(42, 4096)


def frame(k):
    conn.frame_data.jitter, conn.frame_data.progressive_count = (float(jit[k % 4096, 0]), float(jit[k % 4096, 1])), k
    conn.draw()


rates = {}
for mode, depth in (("render only", 0), ("asynchronous presentation, 1 frame ahead", 1), ("asynchronous presentation, 2 frames ahead", 2),
                    ("asynchronous presentation, 3 frames ahead", 3)):
    slots = depth + 1  # the host shows frame k - depth while frames k - depth + 1 .. k are rendered and copied
    for rep in range(3):
        dev.sync()
        t0 = time.perf_counter()
        for k in range(FRAMES):
            frame(k)
            if depth:
                target.present_async(k % slots)
                if k >= depth:
                    px = target.present_wait((k - depth) % slots, copy=False)
        for k in range(max(FRAMES - depth, 0), FRAMES):
            if depth:
                px = target.present_wait(k % slots, copy=False)
        dev.sync()
        dt = (time.perf_counter() - t0) / FRAMES
    st = dev.stats()
    rates[mode] = st.last_segments / dt / 1e9
    print("%-44s %.3f ms per frame = %.1f Grays/s, %.0f frames/s%s" % (mode + ":", dt * 1e3, rates[mode], 1.0 / dt,
          "  (%.1f GB/s of texels, %.2f of the render-only rate)" % (W * H * 4 / dt / 1e9, rates[mode] / rates["render only"]) if depth else ""))
for upload in (False, True):
    for rep in range(2):
        t0 = time.perf_counter()
        seg = 0
        for k in range(200):
            if upload:
                rt.write(textures[k & 3])
            conn.frame_data.jitter, conn.frame_data.progressive_count = (float(jit[k, 0]), float(jit[k, 1])), k
            conn.draw()
            px = target.read_rgba8()
        dt = (time.perf_counter() - t0) / 200
    st = dev.stats()
    print("render + read-back%s: %.3f ms per frame = %.1f Grays/s, %.0f frames/s (kernel %.3f ms)" % (
        " + random-texture upload" if upload else "", dt * 1e3, st.last_segments / dt / 1e9, 1.0 / dt, st.last_kernel_ms))
dev.close()
