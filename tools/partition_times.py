#!/usr/bin/env python3
"""Kernel time of ONE rank's share of the headline frame under an N-way strip partition, measured on
one GPU (every rank of an N-GPU run executes exactly this) — predicts tile-parallel strong scaling."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cases  # noqa: E402
from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, synth  # noqa: E402
from rsraytracingv3_amd import _native as N  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
scene = cases.headline_scene(8)
dev = Device(0)
dev.set_outputs(N.RT_OUT_RGBA8)
rt = RandomTexture(dev)
rt.write(synth.random_texture(1))
sky = synth.procedural_skybox(2048, 1024)
jit = synth.jitter_sequence(42, 40)
base = None
for world in (1, 2, 4, 8):
    for variant in (-1, 0, 2, 3):
        dev.set_partition(0, world, 8)
        dev.set_variant(variant)
        RenderTarget.new(dev, (W, H))
        conn = Connection(scene, dev, sky, rt, max_objects="live")
        conn.update_buffers((W, H), scene, (0.0, 0.0))
        kms = np.zeros(40, np.float32)
        for rep in range(2):
            for k in range(40):
                conn.frame_data.jitter = (float(jit[k, 0]), float(jit[k, 1]))
                conn.draw()
            dev.call("rt_get_frame_history", 40, None, None, kms.ctypes.data_as(C.c_void_p))
        t = float(np.median(kms))
        if world == 1 and variant == -1:
            base = t
        print("%dx%d  world %d  variant %2d: %.4f ms per rank  -> ideal-overlap speedup %.2fx" % (W, H, world, variant, t, base / t))
