#!/usr/bin/env python3
"""Frames per launch (rt_render_frames) against throughput, in one process on one device:
 (a) the four workloads of DESIGN.md §5 at 1, 2, 4, 8, 16 frames per launch;
 (b) ONE rank's share of the headline frame under an N-way strip partition with 8*N frames per launch
     (what every rank of an N-GPU run executes) — wall time per frame including the resolve pass."""
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

dev = Device(0)
dev.set_outputs(N.RT_OUT_RGBA8)
rt = RandomTexture(dev)
rt.write(synth.random_texture(1))
sky = synth.procedural_skybox(2048, 1024)
jit = synth.jitter_sequence(42, 1024)


def measure(scene, size, batch, frames, world=1):
    dev.set_partition(0, world, 8)
    RenderTarget.new(dev, size)
    conn = Connection(scene, dev, sky, rt, max_objects="live")
    conn.update_buffers(size, scene, (0.0, 0.0))
    best, seg = 1e9, 0
    for rep in range(3):
        dev.sync()
        t0 = time.perf_counter()
        k = 0
        while k < frames:
            if batch == 1:
                conn.frame_data.jitter, conn.frame_data.progressive_count = (float(jit[k, 0]), float(jit[k, 1])), k
                conn.draw()
            else:
                conn.draw_frames(jit[k:k + batch], k)
            k += batch
        dev.sync()
        best = min(best, (time.perf_counter() - t0) / frames)
        st = dev.stats()
        seg = st.last_segments / st.last_batch
    return best, seg


print("(a) Grays/s (wall, incl. resolve) by frames per launch")
work = [("1920x1080, 65 obj, 8 bounces", cases.headline_scene(8), (1920, 1080)),
        ("3840x2160, 65 obj, 8 bounces", cases.headline_scene(8), (3840, 2160)),
        ("1920x1080, 1025 obj, 8 bounces", cases.stress_scene(1024, 8), (1920, 1080)),
        ("1920x1080, 65 obj, 1 bounce", cases.headline_scene(1), (1920, 1080))]
for name, scene, size in work:
    row = []
    for b in (1, 2, 4, 8, 16):
        frames = 64 if "1025" not in name else 32
        t, seg = measure(scene, size, b, frames)
        row.append("%d: %.2f" % (b, seg / t / 1e9))
    print("  %-32s %s" % (name, "   ".join(row)), flush=True)
print("(b) one rank's share of the headline frame, 8*N frames per launch: ms per frame (wall)")
for size in ((1920, 1080), (3840, 2160)):
    base = None
    for world in (1, 2, 4, 8):
        b = 8 * world
        t, _ = measure(cases.headline_scene(8), size, b, 4 * b, world)
        t1, _ = measure(cases.headline_scene(8), size, 1, 64, world)
        base = base or t
        print("  %dx%d  N=%d: %.4f ms (ideal-gather speedup %.2fx);  one frame per launch: %.4f ms" % (size[0], size[1], world, t * 1e3, base / t, t1 * 1e3), flush=True)
dev.close()
