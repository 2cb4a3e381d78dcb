#!/usr/bin/env python3
"""PCIe-inclusive frame rate at the C-ABI boundary: render one 1080p frame, read its RGBA8 texels back to
(pageable) host memory, every frame — what the reference's loop does when it shows each frame — with and
without a fresh 600x600 random texture uploaded per frame (random_texture.rs:56-75 does so every 4 ms)."""
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
jit = synth.jitter_sequence(42, 256)
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
