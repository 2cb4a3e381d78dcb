#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: 3840x2160, 64 spheres, 8 bounces, 256 accumulated frames against an
8192x4096 RGBA32F skybox (512 MiB, larger than the 256 MiB Infinity Cache).  Reports the frame rate and
Mrays/s of the whole accumulation and compares the 2k-skybox run (is the texture path visible at all?)."""
import ctypes as C
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

W, H, FRAMES = 3840, 2160, 256
scene = cases.headline_scene(8)
dev = Device(0)
dev.set_outputs(N.RT_OUT_RGBA8)
jit = synth.jitter_sequence(5, FRAMES)
for sky_w in (2048, 8192):
    sky = synth.procedural_skybox(sky_w, sky_w // 2)
    for mode, name in ((N.RT_ACCUM_SRGB8, "sRGB8 feedback (reference)"), (N.RT_ACCUM_FLOAT32, "float32 history")):
        dev.set_accum_mode(mode)
        target = RenderTarget.new(dev, (W, H))
        rt = RandomTexture(dev, seed=100)
        rt.generate_on_device()
        conn = Connection(scene, dev, sky, rt, max_objects="live")
        conn.update_buffers((W, H), scene, (0.0, 0.0))
        dev.sync()
        for per_launch in (1, 8):  # the reference's loop turn by turn, and the same 256 frames eight per launch
            conn.frame_data.progressive_count = 0
            dev.sync()
            t0 = time.perf_counter()
            for k in range(0, FRAMES, per_launch):
                if k % 8 == 0:
                    rt.generate_on_device()          # the reference refreshes the texture every ~4 ms
                if per_launch == 1:
                    conn.frame_data.jitter = (float(jit[k, 0]), float(jit[k, 1]))
                    conn.frame_data.progressive_count = k
                    conn.draw()
                else:
                    conn.draw_frames(jit[k:k + per_launch], k)
            dev.sync()
            dt = time.perf_counter() - t0
            launches = FRAMES // per_launch
            seg = np.zeros(launches, np.uint64)
            kms = np.zeros(launches, np.float32)
            dev.call("rt_get_frame_history", launches, seg.ctypes.data_as(C.c_void_p), None, kms.ctypes.data_as(C.c_void_p))
            span = C.c_float(0.0)  # launches may overlap (four single frames in flight): device time = first kernel's start .. last one's end
            dev.call("rt_get_launch_span", launches, C.byref(span))
            print("skybox %dx%d (%d MiB), %s, %d frame(s) per launch: %d frames in %.1f ms = %.0f fps, %.1f Grays/s wall, %.1f Grays/s over the "
                  "render kernels' span" % (sky_w, sky_w // 2, sky_w * sky_w // 2 * 16 >> 20, name, per_launch, FRAMES, dt * 1e3, FRAMES / dt,
                                            seg.sum() / dt / 1e9, seg.sum() / (span.value * 1e-3) / 1e9), flush=True)
    del sky
