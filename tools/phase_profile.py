#!/usr/bin/env python3
"""Where the waves of the render kernel spend their cycles: runs the profiling build (make -C
rsraytracingv3_amd/csrc prof -> librt355_prof.so, s_memtime stamps between the phases of the render kernels)
on the headline workload and prints each phase's share of the summed wave lifetime.

  RT355_LIB=rsraytracingv3_amd/librt355_prof.so python tools/phase_profile.py [--batch 8] [--variant -1]
The stamps cost ~10 % themselves; use the shares, not the absolute time."""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("RT355_LIB", os.path.join(ROOT, "rsraytracingv3_amd", "librt355_prof.so"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cases  # noqa: E402
from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, synth  # noqa: E402
from rsraytracingv3_amd import _native as N  # noqa: E402

NAMES = ["primary step (culled scan + shade)", "full sphere scan", "cooperative scan (<= 32 rays)", "finish_hit + shade",
         "count barrier (wait for the block)", "re-pack (exchange + 2nd barrier)", "path_end of the last paths",
         "staging + first barrier"]

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--variant", type=int, default=-1)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--spheres", type=int, default=64)
ap.add_argument("--tail-spill", type=int, default=-1, help="RT_OPT_TAIL_SPILL (tail_stream's stamps: 5 refill, 3 shading, 1 scan, 6 misses)")
ap.add_argument("--refill-min", type=int, default=1)
ap.add_argument("--slots", type=int, default=0, help="80: the reference's buffer with its zero-filled tail instead of the live objects")
args = ap.parse_args()
W, H = args.width, args.height
scene = cases.headline_scene(8) if args.spheres == 64 else cases.stress_scene(args.spheres, 8)
dev = Device(0)
dev.set_outputs(N.RT_OUT_RGBA8)
dev.set_variant(args.variant)
dev.set_option(N.RT_OPT_BATCH_INFLIGHT, 1)  # one launch at a time: the kernel time printed is one kernel's alone
dev.set_option(N.RT_OPT_INFLIGHT, 1)
if args.tail_spill >= 0:
    dev.set_option(N.RT_OPT_TAIL_SPILL, args.tail_spill)
dev.set_option(N.RT_OPT_REFILL_MIN, args.refill_min)
target = RenderTarget.new(dev, (W, H))
rt = RandomTexture(dev)
rt.write(synth.random_texture(1))
conn = Connection(scene, dev, synth.procedural_skybox(2048, 1024), rt, max_objects=(args.slots if args.slots else "live"))
conn.update_buffers((W, H), scene, (0.0, 0.0))
jit = synth.jitter_sequence(42, 4096)
k = 0
words = (C.c_uint64 * 16)()
for it in range(40):
    if it == 39:  # the event counts (words 12..15) run since the previous read: make them the last launch's
        dev.sync()
        dev.call("rt_get_counter_words", words)
    if args.batch == 1:
        conn.frame_data.jitter = (float(jit[k, 0]), float(jit[k, 1]))
        conn.frame_data.progressive_count = k
        conn.draw()
    else:
        conn.draw_frames(jit[k:k + args.batch], k)
    k += args.batch
dev.sync()
dev.call("rt_get_counter_words", words)
st = dev.stats()
ph = np.array(words[4:12], dtype=np.float64)
print("variant %d, %d frame(s) per launch, kernel %.3f ms, %d segments, %d wave iterations"
      % (st.last_variant, args.batch, st.last_kernel_ms, st.last_segments, st.last_wave_iterations))
for name, v in zip(NAMES, ph):
    print("  %-42s %6.2f %%   %.3e wave-cycles" % (name, 100.0 * v / ph.sum(), v))
ev = [int(w) for w in words[12:16]]
if any(ev) and "events" in os.environ.get("RT355_LIB", ""):  # event-count build: words 4..7 are the direction-mask counts
    dmc = [int(w) for w in words[4:8]]
    if dmc[0]:
        print("  direction-mask scans of that launch: %d wave-scans, %.2f loop trips each, %.1f rays each, %.2f candidates per ray"
              % (dmc[0], dmc[1] / dmc[0], dmc[3] / dmc[0], dmc[2] / max(dmc[3], 1)))
if any(ev):
    print("  events of that launch: full scans tested %d spheres per wave-scan total, root + update taken for %d of them "
          "(%.1f %%); cooperative scans %d sphere steps (candidate lists: rounds); listed (primary) scans %d spheres"
          % (ev[0], ev[1], 100.0 * ev[1] / max(ev[0], 1), ev[2], ev[3]))
