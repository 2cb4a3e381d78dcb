#!/usr/bin/env python3
"""Generates tests/golden/*.npz with the CPU oracle (oracle/rt_oracle.c).

The reference cannot be built or run here and ships no vectors, so these
fixtures are outputs of OUR restatement — they freeze the arithmetic specification so that
neither the oracle nor the HIP path can drift silently; they are not reference outputs.
Inputs are regenerated from seeds by tests/cases.py + rsraytracingv3_amd/synth.py; a CRC of
every input array is stored beside the outputs.   Run:  python tests/golden/make_golden.py
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import cases  # noqa: E402
import oracle  # noqa: E402

# name -> (scene factory, (W, H), skybox (w, h), jitter, object slots, frames)
CASES = {
    "config1_64": (cases.config1_scene, (64, 64), (64, 32), [(0.25, -0.125)], None),
    "headline_96x54_r8": (lambda: cases.headline_scene(8), (96, 54), (256, 128), [(-0.3, 0.41)], None),
    "horizon_98x88_r3": (lambda: cases.horizon_scene(3), (98, 88), (256, 128), [(0.0, 0.0)], None),
    "accumulate_64x36_r3": (lambda: cases.headline_scene(3), (64, 36), (128, 64),
                            [(0.1, -0.2), (-0.33, 0.27), (0.45, 0.05)], None),
    "stress1024_48x27_r8": (lambda: cases.stress_scene(1024, 8), (48, 27), (128, 64), [(0.2, 0.3)], "live"),
}


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def render_case(name):
    factory, size, sky_size, jitters, slots = CASES[name]
    scene = factory()
    sky, rnd = cases.inputs(*sky_size)
    w, h = size
    if slots == "live":
        objects, count = scene.live_object_bytes(), len(scene.objects)
    else:
        objects, count = scene.as_bytes(w, h)[0], 80
    cfg = scene.config_bytes(w, h)
    out, history = {}, None
    for k, j in enumerate(jitters):
        f = oracle.render(objects, count, cfg, j, k, sky, rnd, history_rgba8=history)
        out["rgbf32_%d" % k], out["rgba8_%d" % k], out["segments_%d" % k] = f.rgbf32, f.rgba8, f.segments.astype(np.uint8)
        history = f.rgba8
    out["input_crc"] = np.array([crc(np.frombuffer(objects, np.uint8)), crc(np.frombuffer(cfg, np.uint8)), crc(sky),
                                 crc(rnd)], np.uint32)
    return out, dict(scene=scene, size=size, sky=sky, rnd=rnd, jitters=jitters, slots=slots)


if __name__ == "__main__":
    for name in CASES:
        out, _ = render_case(name)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, {k: v.shape for k, v in out.items()})
