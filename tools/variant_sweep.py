#!/usr/bin/env python3
"""Which launch shape wins for one rank's share of the headline frame (N-way strip partition, measured on ONE GPU):
wall-clock ms per frame for variant 0 (plain tiles) and 3 (re-packing workgroups), one frame per launch (launches
in flight) and 8*N frames per launch.  Feeds the auto rule of launch_render."""
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
jit = synth.jitter_sequence(42, 2048)
scene = cases.headline_scene(8)
for size in ((1920, 1080), (3840, 2160)):
    for world in (1, 2, 4, 8):
        row = []
        for variant in (0, 3):
            dev.set_variant(variant)
            dev.set_partition(0, world, 8)
            RenderTarget.new(dev, size)
            conn = Connection(scene, dev, sky, rt, max_objects="live")
            conn.update_buffers(size, scene, (0.0, 0.0))
            res = []
            for batch in (1, 8 * world):
                frames = 256 if batch == 1 else 4 * batch
                best = 1e9
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
                res.append(best * 1e3)
            row.append("variant %d: %.4f / %.4f" % (variant, res[0], res[1]))
        print("%dx%d N=%d  ms per frame, single / batched:  %s" % (size[0], size[1], world, "   ".join(row)), flush=True)
dev.set_variant(-1)
dev.set_partition(0, 1, 8)
dev.close()
