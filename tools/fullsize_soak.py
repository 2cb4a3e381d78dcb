#!/usr/bin/env python3
"""Full-size soak: N progressive 1920x1080 frames of the headline scene rendered as 8-frame launches, every
launch's texels and floats compared bit for bit with the oracle's frame chain (and the segment totals)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import cases  # noqa: E402
import oracle  # noqa: E402
from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, synth  # noqa: E402
from rsraytracingv3_amd import _native as N  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 3
W, H, B = 1920, 1080, 8
scene = cases.headline_scene(8)
sky, rnd = synth.procedural_skybox(2048, 1024), synth.random_texture(1)
jit = synth.jitter_sequence(123, launches * B)
dev = Device(0)
dev.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32)
target = RenderTarget.new(dev, (W, H))
rt = RandomTexture(dev)
rt.write(rnd)
conn = Connection(scene, dev, sky, rt, max_objects="live")
conn.update_buffers((W, H), scene, (0.0, 0.0))
objects, count, cfg = scene.live_object_bytes(), len(scene.objects), scene.config_bytes(W, H)
threads = min(oracle.max_threads(), os.cpu_count() or 1, 16)
history, bad = None, 0
for b in range(launches):
    conn.draw_frames(jit[b * B:(b + 1) * B], b * B)
    dev.sync()
    st = dev.stats()
    seg = 0
    for k in range(b * B, (b + 1) * B):
        ref = oracle.render(objects, count, cfg, (float(jit[k, 0]), float(jit[k, 1])), k, sky, rnd, history_rgba8=history,
                            threads=threads, want_hits=False)
        history = ref.rgba8
        seg += int(ref.segments.sum())
    ok = (np.array_equal(target.read_rgba8(), ref.rgba8)
          and np.array_equal(target.read_rgbf32().view(np.uint32), ref.rgbf32.view(np.uint32)) and st.last_segments == seg)
    bad += not ok
    print("launch %d (frames %d..%d): %s, %d segments" % (b, b * B, b * B + B - 1, "bit-identical" if ok else "MISMATCH", seg), flush=True)
dev.close()
print("full-size soak: %d frames, %d mismatching launches" % (launches * B, bad))
sys.exit(1 if bad else 0)
