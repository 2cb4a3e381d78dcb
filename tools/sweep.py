#!/usr/bin/env python3
"""Interleaved A/B timing of kernel options in ONE process on ONE device (devices differ by
up to ~12 %, so numbers from different gpurun boxes must not be compared).

  python tools/sweep.py --opt variant=0,2 --opt blocks_per_cu=3,4,5 --rounds 5 --frames 40
prints the median/min HIP-event kernel time per combination."""
import argparse
import ctypes as C
import itertools
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

OPTS = {"variant": 0, "refill_min": 1, "blocks_per_cu": 2, "tile_chunk": 3, "flags": 4}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--opt", action="append", default=[])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--frames", type=int, default=30)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spheres", type=int, default=64)
    ap.add_argument("--bounces", type=int, default=8)
    args = ap.parse_args()
    import cases
    from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, synth
    from rsraytracingv3_amd import _native as N
    names, values = [], []
    for o in args.opt:
        k, v = o.split("=")
        names.append(k)
        values.append([int(x) for x in v.split(",")])
    combos = list(itertools.product(*values)) if names else [()]
    W, H = args.width, args.height
    scene = cases.spheres_scene(args.spheres, args.bounces, request=max(100, int(args.spheres * 1.6))) \
        if args.spheres <= 64 else cases.stress_scene(args.spheres, args.bounces)
    dev = Device(0)
    dev.set_outputs(N.RT_OUT_RGBA8)
    RenderTarget.new(dev, (W, H))
    rt = RandomTexture(dev)
    rt.write(synth.random_texture(1))
    conn = Connection(scene, dev, synth.procedural_skybox(2048, 1024), rt, max_objects="live")
    conn.update_buffers((W, H), scene, (0.0, 0.0))
    jit = synth.jitter_sequence(42, args.frames)
    res = {c: [] for c in combos}
    seg = np.zeros(args.frames, np.uint64)
    kms = np.zeros(args.frames, np.float32)
    for rnd in range(args.rounds + 1):
        for c in combos:
            for k, v in zip(names, c):
                dev.set_option(OPTS[k], v)
            for f in range(args.frames):
                conn.frame_data.jitter = (float(jit[f, 0]), float(jit[f, 1]))
                conn.frame_data.progressive_count = 0
                conn.draw()
            dev.call("rt_get_frame_history", args.frames, seg.ctypes.data_as(C.c_void_p), None,
                     kms.ctypes.data_as(C.c_void_p))
            if rnd > 0:  # round 0 is warm-up
                st = dev.stats()
                occ = st.last_segments / (64.0 * st.last_wave_iterations) if st.last_wave_iterations else 0
                res[c].append((float(np.median(kms)), float(seg.sum()) / float(kms.sum()) / 1e3, occ))
    print("%-40s %10s %10s %10s %8s" % (" ".join(names), "med ms", "min ms", "Mrays/s", "occ"))
    for c in combos:
        r = np.array(res[c])
        print("%-40s %10.4f %10.4f %10.1f %8.3f" % (" ".join(map(str, c)), np.median(r[:, 0]), r[:, 0].min(),
                                                  np.median(r[:, 1]), np.median(r[:, 2])))


if __name__ == "__main__":
    main()
