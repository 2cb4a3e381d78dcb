#!/usr/bin/env python3
"""Wall-clock frame rate of ONE rank's share (N-way strip partition, no gather) with 1, 2 or 3 independent
rt355 contexts rendering alternate frames on their own streams — does frames-in-flight hide the latency
floor of small per-rank shares?"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases  # noqa: E402
from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, synth  # noqa: E402
from rsraytracingv3_amd import _native as N  # noqa: E402

W, H = 1920, 1080
scene = cases.headline_scene(8)
sky = synth.procedural_skybox(2048, 1024)
rnd = synth.random_texture(1)
jit = synth.jitter_sequence(42, 64)
FR = 600
for world in (1, 2, 4, 8):
    for nctx in (1, 2, 3):
        devs, conns = [], []
        for i in range(nctx):
            d = Device(0)
            d.set_outputs(N.RT_OUT_RGBA8)
            d.set_partition(0, world, 8)
            RenderTarget.new(d, (W, H))
            r = RandomTexture(d)
            r.write(rnd)
            c = Connection(scene, d, sky, r, max_objects="live")
            c.update_buffers((W, H), scene, (0.0, 0.0))
            devs.append(d)
            conns.append(c)
        render = [d._lib.rt_render for d in devs]
        handles = [d.handle for d in devs]
        best = 1e9
        for rep in range(3):
            for d in devs:
                d.sync()
            t0 = time.perf_counter()
            for k in range(FR):
                i = k % nctx
                render[i](handles[i], float(jit[k & 63, 0]), float(jit[k & 63, 1]), 0)
            for d in devs:
                d.sync()
            best = min(best, (time.perf_counter() - t0) / FR)
        print("world %d  contexts %d: %.4f ms per frame (wall)" % (world, nctx, best * 1e3), flush=True)
        for d in devs:
            d.close()
