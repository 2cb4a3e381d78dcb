"""GPU edge cases through the C-ABI: empty and degenerate scenes, size extremes, error behaviour,
the concurrent random-texture thread (random_texture.rs:56-61)."""
import ctypes as C
import os
import threading

import numpy as np
import pytest

import cases
import oracle
from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, render_scene, synth
from rsraytracingv3_amd import _native as N
from rsraytracingv3_amd.scene import Camera, Geometry, Material, Object, Scene

pytestmark = pytest.mark.gpu


def frame(device, scene, size, sky, rnd, jitter=(0.1, 0.2), slots=None, variant=-1):
    device.set_partition(0, 1, 8)
    device.set_accum_mode(N.RT_ACCUM_SRGB8)
    device.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32 | N.RT_OUT_SEGMENTS)
    device.set_variant(variant)
    target = RenderTarget.new(device, size)
    rt = RandomTexture(device)
    rt.write(rnd)
    conn = Connection(scene, device, sky, rt, max_objects=slots)
    render_scene(device, conn, target, scene, jitter)
    device.sync()
    return target.read_rgbf32(), target.read_rgba8(), target.read_segments()


def reference(scene, size, sky, rnd, jitter=(0.1, 0.2), slots=None):
    w, h = size
    if slots == "live":
        objects, count = scene.live_object_bytes(), len(scene.objects)
    else:
        objects, count = scene.as_bytes(w, h)[0], 80
    return oracle.render(objects, count, scene.config_bytes(w, h), jitter, 0, sky, rnd)


def check(device, scene, size, slots=None, variant=-1):
    sky, rnd = cases.inputs(64, 32)
    f, b, s = frame(device, scene, size, sky, rnd, slots=slots, variant=variant)
    ref = reference(scene, size, sky, rnd, slots=slots)
    assert np.array_equal(s, ref.segments)
    assert np.array_equal(f.view(np.uint32), ref.rgbf32.view(np.uint32))
    assert np.array_equal(b, ref.rgba8)
    return ref


@pytest.mark.parametrize("variant", [0, 2, 3])
def test_empty_scene_is_all_sky(device, variant):
    scene = Scene(objects=[], camera=Camera((0, 1, 0), (0.2, 0.4, 0.0), 60.0), reflection_limit=5)
    ref = check(device, scene, (70, 41), slots="live", variant=variant)
    assert (ref.segments == 1).all()


@pytest.mark.parametrize("variant", [0, 2, 3])
def test_planes_only_and_unknown_geometry_option(device, variant):
    floor = Object("floor", Material((0.8, 0.8, 0.8), (0, 0, 0), 0, 0.2, 0.3), Geometry.plane((0, 0, 0), (0, 1, 0), 50.0))
    wall = Object("wall", Material((0.2, 0.4, 0.9), (1, 1, 1), 0.5, 0.2, 0.0), Geometry.plane((0, 5, 20), (0, 0, -1), 10.0))
    ghost = Object("ghost", Material(), Geometry(7, (0.0, 1.0, 5.0), radius=3.0))  # option 7: never intersects (ray.hlsl:53,69)
    scene = Scene(objects=[floor, ghost, wall], camera=Camera((0, 4, -10), (0.1, 0.0, 0.0), 70.0), reflection_limit=6)
    check(device, scene, (96, 54), slots="live", variant=variant)


@pytest.mark.parametrize("size", [(1, 1), (1, 37), (53, 1), (9, 7), (8, 8), (17, 64)])
def test_tiny_and_thin_frames(device, size):
    check(device, cases.headline_scene(4), size)


def test_reflection_limit_zero_and_huge(device):
    scene = cases.headline_scene(0)
    sky, rnd = cases.inputs(64, 32)
    f, b, s = frame(device, scene, (40, 24), sky, rnd)
    assert (s == 0).all() and (f == 0).all() and (b[..., :3] == 0).all() and (b[..., 3] == 255).all()
    deep = cases.headline_scene(100)   # > 63: the packed path state of variant 3 cannot hold it -> falls back
    for variant in (-1, 2, 3):
        check(device, deep, (48, 27), variant=variant)


def test_identical_spheres_tie_goes_to_lowest_index(device):
    a = Object("a", Material((1, 0, 0), (1, 0, 0), 2.0, 0, 0.0), Geometry.sphere((0, 0, 8), 2.0))
    b = Object("b", Material((0, 1, 0), (0, 1, 0), 2.0, 0, 0.0), Geometry.sphere((0, 0, 8), 2.0))
    scene = Scene(objects=[a, b], camera=Camera((0, 0, 0), (0, np.pi, 0), 40.0), reflection_limit=1)
    sky, rnd = cases.inputs(64, 32)
    device.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32 | N.RT_OUT_SEGMENTS | N.RT_OUT_HITS)
    target = RenderTarget.new(device, (16, 16))
    rt = RandomTexture(device)
    rt.write(rnd)
    conn = Connection(scene, device, sky, rt, max_objects="live")
    render_scene(device, conn, target, scene, (0, 0))
    device.sync()
    hits = target.read_hits(1)
    assert set(np.unique(hits)) <= {0, -1} and (hits == 0).any()   # never object 1 (ray.hlsl:62 strict '<')


def test_errors_are_reported_not_thrown():
    dev = Device(0)
    try:
        with pytest.raises(N.RtError) as e:
            dev.call("rt_render", 0.0, 0.0, 0)
        assert e.value.code == -3 and "skybox" in str(e.value)          # RT_ERR_NOT_READY
        with pytest.raises(N.RtError) as e:
            dev.call("rt_resize", 0, 10)
        assert e.value.code == -1
        sky = synth.procedural_skybox(16, 8)
        dev.call("rt_set_skybox", 16, 8, sky.ctypes.data_as(C.c_void_p))
        scene = cases.config1_scene()
        dev.call("rt_set_scene", scene.live_object_bytes(), len(scene.objects), None, 0, scene.config_bytes(32, 32))
        dev.call("rt_resize", 64, 32)
        rnd = synth.random_texture(1)
        dev._lib.rt_set_random(dev.handle, 600, 600, rnd.ctypes.data_as(C.c_void_p))
        with pytest.raises(N.RtError) as e:
            dev.call("rt_render", 0.0, 0.0, 0)                           # Config says 32x32, target is 64x32
        assert e.value.code == -1 and "Config" in str(e.value)
        with pytest.raises(N.RtError):
            buf = np.zeros(5, np.uint8)
            dev.call("rt_read_rgba8", buf.ctypes.data_as(C.c_void_p), buf.nbytes)  # wrong size
        assert dev._lib.rt_destroy(None) == -1 and dev._lib.rt_render(None, 0.0, 0.0, 0) == -1
    finally:
        dev.close()


def test_resize_keeps_working_and_restarts_history(device):
    scene = cases.headline_scene(3)
    sky, rnd = cases.inputs(64, 32)
    device.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32)
    device.set_variant(-1)
    target = RenderTarget.new(device, (64, 40))
    rt = RandomTexture(device)
    rt.write(rnd)
    conn = Connection(scene, device, sky, rt)
    for _ in range(3):
        render_scene(device, conn, target, scene, (0.1, 0.1))
    assert conn.frame_data.progressive_count == 2
    target.resize(device, (48, 30))                        # ui.rs:105
    render_scene(device, conn, target, scene, (0.1, 0.1))  # size changed -> dirty -> progressive_count 0
    device.sync()
    assert conn.frame_data.progressive_count == 0
    ref = reference(scene, (48, 30), sky, rnd, (0.1, 0.1))
    assert np.array_equal(target.read_rgbf32().view(np.uint32), ref.rgbf32.view(np.uint32))


def test_random_texture_thread_never_tears_a_frame(device):
    """rt_set_random from another thread while frames render (random_texture.rs:56-61): every frame
    must equal the oracle's frame for ONE of the two textures, never a mix."""
    scene = cases.headline_scene(6)
    sky = synth.procedural_skybox(64, 32)
    tex = [synth.random_texture(11), synth.random_texture(12)]
    size = (96, 54)
    refs = [reference(scene, size, sky, t, (0.2, -0.3)).rgba8 for t in tex]
    assert not np.array_equal(refs[0], refs[1])
    device.set_outputs(N.RT_OUT_RGBA8)
    device.set_variant(-1)
    target = RenderTarget.new(device, size)
    rt = RandomTexture(device)
    rt.write(tex[0])
    conn = Connection(scene, device, sky, rt)
    conn.update_buffers(size, scene, (0.2, -0.3))
    stop = threading.Event()
    errors = []

    def writer():
        k = 0
        while not stop.is_set():
            k ^= 1
            rc = device._lib.rt_set_random(device.handle, 600, 600, tex[k].ctypes.data_as(C.c_void_p))
            if rc != 0:
                errors.append(rc)

    th = threading.Thread(target=writer)
    th.start()
    try:
        import time
        seen = [0, 0]
        deadline = time.time() + 20.0
        frames = 0
        while (min(seen) < 5 or frames < 60) and time.time() < deadline:
            conn.frame_data.progressive_count = 0
            conn.draw()
            got = target.read_rgba8()
            which = [np.array_equal(got, r) for r in refs]
            assert any(which), "frame matches neither texture: torn upload"
            seen[which.index(True)] += 1
            frames += 1
            time.sleep(0.0005)   # let the writer thread have the interpreter
    finally:
        stop.set()
        th.join()
    assert not errors and min(seen) > 0   # both textures were actually seen


def test_device_side_random_texture_matches_host_generator(device):
    """rt_generate_random fills the texture on the GPU with the same SplitMix64 stream the host
    generator (synth.random_texture) uploads — bit for bit, so frames do not change."""
    rt = RandomTexture(device, seed=77)
    rt.generate_on_device()
    assert np.array_equal(rt.read().view(np.uint32), synth.random_texture(77).view(np.uint32))
    rt.generate_on_device()
    assert np.array_equal(rt.read().view(np.uint32), synth.random_texture(78).view(np.uint32))
    scene = cases.headline_scene(5)
    sky = synth.procedural_skybox(64, 32)
    device.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32)
    target = RenderTarget.new(device, (80, 45))
    conn = Connection(scene, device, sky, rt)
    render_scene(device, conn, target, scene, (0.3, 0.1))
    device.sync()
    ref = reference(scene, (80, 45), sky, synth.random_texture(78), (0.3, 0.1))
    assert np.array_equal(target.read_rgbf32().view(np.uint32), ref.rgbf32.view(np.uint32))


def test_scene_too_big_for_lds_falls_back_to_global_scan(device):
    """4500 spheres = 72 KB of records: more than a workgroup's LDS, every shape must still agree."""
    scene = cases.spheres_scene(4500, 4, seed=5, min_r=0.5, max_r=1.5, placement=260.0, request=4600)
    sky, rnd = cases.inputs(64, 32)
    ref = reference(scene, (40, 24), sky, rnd, slots="live")
    for variant in (-1, 2, 3):
        f, b, s = frame(device, scene, (40, 24), sky, rnd, slots="live", variant=variant)
        assert np.array_equal(f.view(np.uint32), ref.rgbf32.view(np.uint32)), variant


def test_two_contexts_on_one_gpu_do_not_interfere(device):
    """Two rt_contexts (own streams, own buffers) rendering different scenes back to back."""
    other = Device(0)
    try:
        sky, rnd = cases.inputs(64, 32)
        a_scene, b_scene = cases.headline_scene(5), cases.config1_scene()
        b_scene.reflection_limit = 4
        setups = []
        for dev, scene, size in ((device, a_scene, (96, 56)), (other, b_scene, (72, 40))):
            dev.set_partition(0, 1, 8)
            dev.set_accum_mode(N.RT_ACCUM_SRGB8)
            dev.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32)
            dev.set_variant(-1)
            target = RenderTarget.new(dev, size)
            rt = RandomTexture(dev)
            rt.write(rnd)
            conn = Connection(scene, dev, sky, rt)
            setups.append((dev, scene, size, target, conn))
        for _ in range(3):            # interleave launches of both contexts without syncing in between
            for dev, scene, size, target, conn in setups:
                conn.update_buffers(size, scene, (0.2, 0.2))
                conn.frame_data.progressive_count = 0
                conn.draw()
        for dev, scene, size, target, conn in setups:
            dev.sync()
            ref = reference(scene, size, sky, rnd, (0.2, 0.2))
            assert np.array_equal(target.read_rgbf32().view(np.uint32), ref.rgbf32.view(np.uint32))
    finally:
        other.close()


def test_destroy_with_frames_in_flight():
    dev = Device(0)
    scene = cases.headline_scene(8)
    sky, rnd = cases.inputs(64, 32)
    dev.set_outputs(N.RT_OUT_RGBA8)
    RenderTarget.new(dev, (640, 360))
    rt = RandomTexture(dev)
    rt.write(rnd)
    conn = Connection(scene, dev, sky, rt)
    conn.update_buffers((640, 360), scene, (0.0, 0.0))
    for _ in range(20):
        conn.draw()
    dev.close()     # rt_destroy waits for the stream before freeing anything


def test_differential_fuzz_against_oracle():
    """tools/fuzz_parity.py: random scenes / cameras / sizes / bounce limits / kernel shapes, all four outputs
    bit-identical to the oracle and the culling self-check silent; every second case also renders 2-6 frames in
    one launch against the oracle frame chain (22 800 cases were run when this was written)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "fuzz_parity", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.run(cases=120, first=10_000) == 0


def test_skybox_from_an_exr_file(device, tmp_path):
    """Connection.load_hdri with a path (connection.rs:176-215 reads ./assets/table_mountain_1_8k.exr) and
    rt_set_skybox_exr: the frame equals the one rendered from the decoded array."""
    import exr_writer as X
    sky = synth.procedural_skybox(128, 64)
    path = str(tmp_path / "sky.exr")
    X.write_exr(path, {"R": (sky[..., 0], "half"), "G": (sky[..., 1], "half"), "B": (sky[..., 2], "half")}, X.PIZ)
    (w, h), decoded = Connection.load_image(path)
    assert (w, h) == (128, 64) and np.array_equal(decoded[..., :3], sky[..., :3].astype(np.float16).astype(np.float32))
    scene = cases.headline_scene(4)
    rnd = synth.random_texture(2)
    frames = []
    for mode, source in (("array", decoded), ("path", path), ("c-abi", decoded)):
        device.set_partition(0, 1, 8)
        device.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32)
        target = RenderTarget.new(device, (96, 64))
        rt = RandomTexture(device)
        rt.write(rnd)
        conn = Connection(scene, device, source, rt)
        if mode == "c-abi":
            device.call("rt_set_skybox", 1, 1, np.zeros(4, np.float32).ctypes.data_as(C.c_void_p))  # something else first
            device.call("rt_set_skybox_exr", path.encode())
        render_scene(device, conn, target, scene, (0.2, -0.3))
        device.sync()
        frames.append(target.read_rgbf32().copy())
    assert np.array_equal(frames[0].view(np.uint32), frames[1].view(np.uint32))
    assert np.array_equal(frames[0].view(np.uint32), frames[2].view(np.uint32))


def test_size_limits_are_refused_not_overflowed(device):
    """Pixel indices are 32-bit on the device: sizes and launches beyond that are errors, never wrap-arounds."""
    with pytest.raises(N.RtError, match="beyond 65536"):
        device.call("rt_resize", 70000, 8)
    with pytest.raises(N.RtError, match="beyond 65536"):
        device.call("rt_resize", 65536, 65536)
    device.call("rt_resize", 64, 64)  # the context stays usable
