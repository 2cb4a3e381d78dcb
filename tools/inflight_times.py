#!/usr/bin/env python3
"""One rank's share of a tile-split frame, one frame per launch, by launches in flight (RT_OPT_INFLIGHT): ms per
frame (wall clock incl. the resolve pass) on ONE GPU — every rank of an N-GPU run executes exactly this."""
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
dev.set_option(N.RT_OPT_TIMING, 0 if "--timing" not in sys.argv else 1)  # a render loop does not ask for kernel times: no timing-only events
rt = RandomTexture(dev)
rt.write(synth.random_texture(1))
sky = synth.procedural_skybox(2048, 1024)
jit = synth.jitter_sequence(42, 1024)
scene = cases.headline_scene(8)
for size in ((1920, 1080), (3840, 2160)):
    for world in (1, 2, 4, 8):
        row = []
        for k in (1, 2, 4, 5, 8, 0):  # 0 = the library's rule
            dev.set_option(N.RT_OPT_INFLIGHT, k)
            dev.set_partition(0, world, 8)
            RenderTarget.new(dev, size)
            conn = Connection(scene, dev, sky, rt, max_objects="live")
            conn.update_buffers(size, scene, (0.0, 0.0))
            best = 1e9
            for rep in range(3):
                dev.sync()
                t0 = time.perf_counter()
                fd = [(float(jit[f, 0]), float(jit[f, 1])) for f in range(256)]
                render = dev._lib.rt_render
                for f in range(256):  # the raw C-ABI call, as a compiled host makes it (Connection.draw adds ~4 us of Python per frame)
                    render(dev.handle, fd[f][0], fd[f][1], f)
                dev.sync()
                best = min(best, (time.perf_counter() - t0) / 256)
            row.append(("%d in flight: %.4f ms" % (k, best * 1e3)) if k else ("the library's rule (%d): %.4f ms" % (dev.stats().last_launches_in_flight, best * 1e3)))
        print("%dx%d  N=%d   %s" % (size[0], size[1], world, "   ".join(row)), flush=True)
dev.set_option(N.RT_OPT_INFLIGHT, 0)
dev.set_partition(0, 1, 8)
dev.close()
