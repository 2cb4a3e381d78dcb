#!/usr/bin/env python3
"""Single-frame launches in flight (RT_OPT_INFLIGHT 3 .. 6) at 1920x1080: ms per frame of the raw C-ABI call for the headline frame, a one-bounce
frame and the 1 024-sphere scene.  RT355_STREAM_PAD=k in the environment creates k idle streams before the context's render streams (rotates the
runtime's deal of streams to hardware queues): profiles/r04_single_frames_in_flight.txt."""
import os, sys, time
ROOT = os.getcwd()
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, synth
from rsraytracingv3_amd import _native as N
dev = Device(0); dev.set_outputs(N.RT_OUT_RGBA8); dev.set_option(N.RT_OPT_TIMING, 0)
rt = RandomTexture(dev); rt.write(synth.random_texture(1))
sky = synth.procedural_skybox(2048, 1024); jit = synth.jitter_sequence(42, 1024)
for label, scene in (("65 obj 8 bounces", cases.headline_scene(8)), ("65 obj 1 bounce", cases.headline_scene(1)), ("1025 obj", cases.stress_scene(1024, 8))):
    size = (1920, 1080)
    row = []
    for k in (3, 4, 5, 6):
        dev.set_option(N.RT_OPT_INFLIGHT, k)
        RenderTarget.new(dev, size)
        conn = Connection(scene, dev, sky, rt, max_objects="live"); conn.update_buffers(size, scene, (0.0, 0.0))
        best = 1e9
        for rep in range(4):
            dev.sync(); t0 = time.perf_counter()
            render = dev._lib.rt_render
            for f in range(480):
                render(dev.handle, float(jit[f, 0]), float(jit[f, 1]), f)
            dev.sync(); best = min(best, (time.perf_counter() - t0) / 480)
        row.append("%d: %.4f" % (k, best * 1e3))
    print(label, "  ms per frame by launches in flight  ", "   ".join(row), flush=True)
