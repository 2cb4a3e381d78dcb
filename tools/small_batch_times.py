#!/usr/bin/env python3
"""Multi-frame launches of FEW frames (rt_render_frames with 2 .. 8 frames) by launches in flight (RT_OPT_BATCH_INFLIGHT):
Grays/s (wall clock incl. the resolve pass) of the headline scene and of the 1 024-sphere scene at 1920x1080.
python tools/small_batch_times.py [--default] [--lift]   (--default: leave the option alone, i.e. measure the library's own
rule; --lift: RT_OPT_FLAGS bit 2, the tail spill whatever the size of the launch)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, synth  # noqa: E402
from rsraytracingv3_amd import _native as N  # noqa: E402

dev = Device(0)
dev.set_outputs(N.RT_OUT_RGBA8)
rt = RandomTexture(dev)
rt.write(synth.random_texture(1))
sky = synth.procedural_skybox(2048, 1024)
jit = synth.jitter_sequence(42, 1024)
size = (1920, 1080)
FRAMES = 240


def measure(conn, batch):
    best, seg = 1e9, 0
    for rep in range(3):
        dev.sync()
        t0 = time.perf_counter()
        k = 0
        while k < FRAMES:
            if batch == 1:
                conn.frame_data.jitter, conn.frame_data.progressive_count = (float(jit[k, 0]), float(jit[k, 1])), k
                conn.draw()
            else:
                conn.draw_frames(jit[k:k + batch], k)
            k += batch
        dev.sync()
        best = min(best, (time.perf_counter() - t0) / FRAMES)
        st = dev.stats()
        seg = st.last_segments / st.last_batch
    return seg / best / 1e9


for label, scene in (("65 objects", cases.headline_scene(8)), ("1025 objects", cases.stress_scene(1024, 8))):
    RenderTarget.new(dev, size)
    conn = Connection(scene, dev, sky, rt, max_objects="live")
    conn.update_buffers(size, scene, (0.0, 0.0))
    print("1920x1080, %s, 8 bounces: Grays/s by frames per launch (rows) and launches in flight (columns)" % label, flush=True)
    print("  one frame per launch (RT_OPT_INFLIGHT default): %.2f" % measure(conn, 1), flush=True)
    for batch in (2, 3, 4, 6, 8):
        if "--lift" in sys.argv:  # RT_OPT_FLAGS bit 2: the tail spill whatever the launch's size
            dev.set_option(N.RT_OPT_FLAGS, 4)
        row = []
        for lanes in ((None,) if "--default" in sys.argv else (1, 2, 3, 4)):
            if lanes is not None:
                dev.set_option(N.RT_OPT_BATCH_INFLIGHT, lanes)
            row.append("%s: %.2f" % ("library's rule" if lanes is None else "%d" % lanes, measure(conn, batch)))
        print("  %d frames per launch   %s" % (batch, "   ".join(row)), flush=True)
dev.close()
