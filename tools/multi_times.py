#!/usr/bin/env python3
"""What the single-process multi-device front end (rt_multi_*) costs on ONE GPU: the headline frame through rt_create,
through an rt_multi that binds the device once, and through rt_multis that bind it 2, 4 and 8 times (every member renders
its strips, the strips are copied and assembled on the first member: all of the N-GPU machinery, sharing one device).
ms per frame, 8 frames per launch and one frame per launch.   python tools/multi_times.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from rsraytracingv3_amd import Connection, Device, MultiDevice, RandomTexture, RenderTarget, synth  # noqa: E402
from rsraytracingv3_amd import _native as N  # noqa: E402

W, H = 1920, 1080
scene = cases.headline_scene(8)
sky = synth.procedural_skybox(2048, 1024)
rnd = synth.random_texture(1)
jit = synth.jitter_sequence(42, 2048)


DRY = "--dry" in sys.argv       # RT_OPT_FLAGS bit 4: every kernel runs one workgroup — the host's calls without the device's work
TIMING = "--timing" in sys.argv  # keep the timing-only events of single-frame launches (RT_OPT_TIMING = 1, the library default)


def measure(dev, label):
    dev.set_outputs(N.RT_OUT_RGBA8)
    dev.set_option(N.RT_OPT_TIMING, 1 if TIMING else 0)
    if DRY:
        dev.set_option(N.RT_OPT_FLAGS, 16)
    RenderTarget.new(dev, (W, H))
    rt = RandomTexture(dev)
    rt.write(rnd)
    conn = Connection(scene, dev, sky, rt, max_objects="live")
    conn.update_buffers((W, H), scene, (0.0, 0.0))
    out = []
    for batch, frames in ((8, 512), (1, 256)):
        best = 1e9
        for rep in range(4):
            dev.sync()
            t0 = time.perf_counter()
            k = 0
            while k < frames:
                if batch == 1:  # the raw C-ABI call, as a compiled host makes it
                    dev._lib.rt_render(dev.handle, float(jit[k, 0]), float(jit[k, 1]), k)
                else:
                    conn.draw_frames(jit[k:k + batch], k)
                k += batch
            dev.sync()
            best = min(best, (time.perf_counter() - t0) / frames)
        out.append(best * 1e3)
    print("%-44s %.4f ms per frame at 8 frames per launch, %.4f at one frame per launch" % (label, out[0], out[1]), flush=True)
    dev.close()


measure(Device(0), "rt_create (one context)")
for n in (1, 2, 4, 8):
    measure(MultiDevice([0] * n, 8), "rt_multi_create, the device bound %d time(s)" % n)
