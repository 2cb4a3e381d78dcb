#!/usr/bin/env python3
"""Single-frame launches in flight (RT_OPT_INFLIGHT 4 against 5) by reflection limit, headline scene at 1920x1080: ms per frame of the raw C-ABI call,
every (limit, lanes) cell measured three times in turn (the cells are bimodal from run to run: all three are printed)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cases
from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, synth
from rsraytracingv3_amd import _native as N
dev = Device(0); dev.set_outputs(N.RT_OUT_RGBA8); dev.set_option(N.RT_OPT_TIMING, 0)
rt = RandomTexture(dev); rt.write(synth.random_texture(1))
sky = synth.procedural_skybox(2048, 1024); jit = synth.jitter_sequence(42, 1024)
size = (1920, 1080)
for bounces in (1, 2, 3, 4, 5, 6, 8, 12):
    scene = cases.headline_scene(bounces)
    cells = {4: [], 5: []}
    for rep in range(3):
        for k in (4, 5):
            dev.set_option(N.RT_OPT_INFLIGHT, k)
            RenderTarget.new(dev, size)
            conn = Connection(scene, dev, sky, rt, max_objects="live"); conn.update_buffers(size, scene, (0.0, 0.0))
            best = 1e9
            for r in range(2):
                dev.sync(); t0 = time.perf_counter()
                render = dev._lib.rt_render
                for f in range(480):
                    render(dev.handle, float(jit[f, 0]), float(jit[f, 1]), f)
                dev.sync(); best = min(best, (time.perf_counter() - t0) / 480)
            cells[k].append(best * 1e3)
    print("%2d bounces   4 in flight: %s   5 in flight: %s" % (bounces, " ".join("%.4f" % v for v in cells[4]), " ".join("%.4f" % v for v in cells[5])), flush=True)
