import os, sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import cases
from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, synth
from rsraytracingv3_amd import _native as N
dev = Device(0); dev.set_outputs(N.RT_OUT_RGBA8)
rt = RandomTexture(dev); rt.write(synth.random_texture(1))
sky = synth.procedural_skybox(256, 128); jit = synth.jitter_sequence(42, 1024)
scene = cases.headline_scene(8)
for size in ((64, 64), (1920, 1080)):
    for world in (1, 8):
        for k in (1, 4):
            dev.set_option(N.RT_OPT_INFLIGHT, k); dev.set_partition(0, world, 8)
            RenderTarget.new(dev, size); conn = Connection(scene, dev, sky, rt, max_objects="live"); conn.update_buffers(size, scene, (0.0, 0.0))
            fd = conn.frame_data.as_bytes(); best = 1e9
            for rep in range(3):
                dev.sync(); t0 = time.perf_counter()
                for f in range(512):
                    dev._lib.rt_render(dev.handle, 0.1, 0.2, f)
                dev.sync(); best = min(best, (time.perf_counter() - t0) / 512)
            print("%dx%d N=%d inflight %d: %.2f us per rt_render call (raw ctypes)" % (size[0], size[1], world, k, best * 1e6), flush=True)
