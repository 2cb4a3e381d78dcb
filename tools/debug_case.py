#!/usr/bin/env python3
"""Debug helper: one fuzz case on the GPU against the oracle, listing the first differing hit sequences.
   python tools/debug_case.py SEED [variant]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np  # noqa: E402
import oracle  # noqa: E402
import fuzz_parity as F  # noqa: E402
from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, render_scene, synth  # noqa: E402
from rsraytracingv3_amd import _native as N  # noqa: E402

seed = int(sys.argv[1])
scene, size, jitter, slots, variant, tex_seed = F.random_case(seed)
if len(sys.argv) > 2:
    variant = int(sys.argv[2])
dev = Device(0)
sky = synth.procedural_skybox(128, 64)
rnd = synth.random_texture(tex_seed)
dev.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32 | N.RT_OUT_SEGMENTS | N.RT_OUT_HITS)
dev.set_variant(variant)
dev.set_option(N.RT_OPT_CULL, 2)
dev.set_option(N.RT_OPT_GRID_MIN, F.grid_min_for(seed))
for opt, val in list(F.dirmask_for(seed).items()) + list(F.dirlist_for(seed).items()):
    dev.set_option(opt, val)
target = RenderTarget.new(dev, size)
rt = RandomTexture(dev)
rt.write(rnd)
conn = Connection(scene, dev, sky, rt, max_objects=slots)
render_scene(dev, conn, target, scene, jitter)
dev.sync()
st = dev.stats()
w, h = size
objects, count = (scene.live_object_bytes(), len(scene.objects)) if slots == "live" else (scene.as_bytes(w, h)[0], 80)
ref = oracle.render(objects, count, scene.config_bytes(w, h), jitter, 0, sky, rnd, threads=8)
got = target.read_hits(scene.reflection_limit)
bad = np.argwhere((got != ref.hits).any(-1))
print("seed %d: %d objects, size %s, R %d, variant used %d stage %d, cull mismatches %d, pixels with differing hit sequences %d; table %d x %d lists %d"
      % (seed, len(scene.objects), size, scene.reflection_limit, st.last_variant, st.last_stage, st.last_cull_mismatches, len(bad), st.dirmask_cells,
         st.dirmask_bins, st.dirmask_lists))
first = {}
for y, x in bad[:4000]:
    b = int(np.argmax(got[y, x] != ref.hits[y, x]))
    key = (b, int(ref.hits[y, x, b]), int(got[y, x, b]), int(ref.hits[y, x, b - 1]) if b else -9)
    first[key] = first.get(key, 0) + 1
for k, v in sorted(first.items(), key=lambda kv: -kv[1])[:25]:
    print("  first difference at bounce %d: expected %d got %d (previous hit %d): %d pixels" % (k[0], k[1], k[2], k[3], v))
