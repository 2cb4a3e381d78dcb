"""GPU parity: the HIP render path (through the C-ABI) against the CPU oracle, bit for bit.

Both sides implement the same written float32 specification independently, so
the bar is exact equality of the linear-float output, the stored sRGB8 texels, the per-pixel
segment counts and the per-segment hit indices.  The 1e-4 tolerance BASELINE.json states is
asserted as well (it is implied)."""
import ctypes as C

import numpy as np
import pytest

import cases
import oracle
from rsraytracingv3_amd import Connection, RandomTexture, RenderTarget, render_scene, synth
from rsraytracingv3_amd import _native as N

pytestmark = pytest.mark.gpu

TOL = 1e-4  # BASELINE.json north_star: per-pixel RGB within 1e-4
# The resident-wave (2) and streamed (4) launch shapes only exist in the A/B build (librt355_ab.so); the shipped library runs the
# re-packing shape for both — every test below holds for either, tests/test_ab_shapes_gpu.py re-runs the shape-specific ones on the A/B build.
AB = bool(N.load().rt_has_ab_shapes())


def run_gpu(device, scene, size, sky, rnd, jitter, max_objects=None, frames=1, variant=-1, accum=N.RT_ACCUM_SRGB8,
            refill_min=16):
    device.set_partition(0, 1, 8)
    device.set_accum_mode(accum)
    device.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32 | N.RT_OUT_SEGMENTS | N.RT_OUT_HITS)
    device.set_variant(variant)
    device.set_option(N.RT_OPT_REFILL_MIN, refill_min)
    target = RenderTarget.new(device, size)
    rt = RandomTexture(device)
    rt.write(rnd)
    conn = Connection(scene, device, sky, rt, max_objects=max_objects)
    outs = []
    for k in range(frames):
        render_scene(device, conn, target, scene, jitter[k] if frames > 1 else jitter)
        device.sync()
        outs.append((target.read_rgbf32(), target.read_rgba8(), target.read_segments(),
                     target.read_hits(scene.reflection_limit), conn.frame_data.progressive_count))
    return outs, device.stats()


def run_oracle(scene, size, sky, rnd, jitter, max_objects=None, pc=0, history=None, history_f32=None, accum=0):
    w, h = size
    if max_objects == "live":
        objects, count = scene.live_object_bytes(), len(scene.objects)
    else:
        slots = 80 if max_objects is None else max_objects
        objects, count = scene.as_bytes(w, h, slots)[0], slots
    return oracle.render(objects, count, scene.config_bytes(w, h), jitter, pc, sky, rnd,
                         history_rgba8=history, history_f32=history_f32, accum_mode=accum)


def assert_frame_equal(gpu, ref):
    rgbf, rgba8, seg, hits, _ = gpu
    assert np.array_equal(seg, ref.segments), "segment counts differ in %d pixels" % np.count_nonzero(seg != ref.segments)
    bad = np.count_nonzero((hits != ref.hits).any(axis=-1))
    assert bad == 0, "hit-index sequences differ in %d pixels" % bad
    diff = np.abs(rgbf.astype(np.float64) - ref.rgbf32.astype(np.float64))
    assert np.nanmax(diff) <= TOL, "max |dRGB| = %g" % np.nanmax(diff)
    assert np.array_equal(rgbf.view(np.uint32), ref.rgbf32.view(np.uint32)), \
        "linear float output not bit-identical (%d words differ)" % np.count_nonzero(
            rgbf.view(np.uint32) != ref.rgbf32.view(np.uint32))
    assert np.array_equal(rgba8, ref.rgba8)


@pytest.mark.parametrize("variant", [-1, 0, 1, 2, 3])
def test_config1_plumbing_256(device, variant):
    """BASELINE config 1: 256x256, 4 spheres + plane, 1 bounce, reference-exact 80 slots."""
    scene = cases.config1_scene()
    sky, rnd = cases.inputs(64, 32)
    outs, st = run_gpu(device, scene, (256, 256), sky, rnd, (0.25, -0.125), variant=variant)
    ref = run_oracle(scene, (256, 256), sky, rnd, (0.25, -0.125))
    assert_frame_equal(outs[0], ref)
    assert st.last_segments == int(ref.segments.sum())
    assert st.sphere_count == 79 and st.plane_count == 1  # 4 live + 75 zero-padded slots (D5)


@pytest.mark.parametrize("variant", [-1, 0, 1, 2, 3])
def test_headline_scene_small(device, variant):
    """BASELINE config 2 scene (64 spheres + plane, 8 bounces) at 320x180."""
    scene = cases.headline_scene(8)
    sky, rnd = cases.inputs(512, 256)
    outs, st = run_gpu(device, scene, (320, 180), sky, rnd, (-0.3, 0.41), variant=variant)
    ref = run_oracle(scene, (320, 180), sky, rnd, (-0.3, 0.41))
    assert_frame_equal(outs[0], ref)
    assert st.last_segments == int(ref.segments.sum())
    assert ref.segments.max() == 8 and ref.segments.min() >= 1


def test_live_count_equals_padded(device):
    """Explicit count 65 and the reference's 80 zero-padded slots give the same image (D5)."""
    scene = cases.headline_scene(4)
    sky, rnd = cases.inputs(256, 128)
    a, _ = run_gpu(device, scene, (200, 120), sky, rnd, (0.1, 0.2), max_objects=None)
    b, _ = run_gpu(device, scene, (200, 120), sky, rnd, (0.1, 0.2), max_objects="live")
    assert np.array_equal(a[0][0].view(np.uint32), b[0][0].view(np.uint32))
    ref = run_oracle(scene, (200, 120), sky, rnd, (0.1, 0.2), max_objects="live")
    assert_frame_equal(b[0], ref)


def test_reference_default_scene_screenshot_view(device):
    """The reference's own start-up scene (77 objects in its 80-slot buffer), its screenshot's camera
    and bounce count, at the screenshot's panel aspect."""
    scene = cases.reference_default_scene(3)
    sky, rnd = cases.inputs(512, 256)
    outs, st = run_gpu(device, scene, (294, 264), sky, rnd, (0.12, -0.34))
    ref = run_oracle(scene, (294, 264), sky, rnd, (0.12, -0.34))
    assert st.object_count == 80 and st.sphere_count == 79 and st.plane_count == 1
    assert_frame_equal(outs[0], ref)


def test_horizon_view(device):
    """Screenshot camera: sky rows exit after one segment, sphere rows bounce."""
    scene = cases.horizon_scene(3)
    sky, rnd = cases.inputs(512, 256)
    outs, _ = run_gpu(device, scene, (294, 264), sky, rnd, (0.0, 0.0))
    ref = run_oracle(scene, (294, 264), sky, rnd, (0.0, 0.0))
    assert_frame_equal(outs[0], ref)
    top = ref.segments[: 264 // 20]
    assert (top == 1).mean() > 0.9  # sky at the top of the frame (D9)


def test_progressive_accumulation_srgb8(device):
    """Three frames of reference-exact accumulation through the sRGB8 history (D7, D13)."""
    scene = cases.headline_scene(3)
    sky, rnd = cases.inputs(256, 128)
    jit = [(0.1, -0.2), (-0.33, 0.27), (0.45, 0.05)]
    outs, _ = run_gpu(device, scene, (160, 96), sky, rnd, jit, frames=3)
    history = None
    for k in range(3):
        assert outs[k][4] == k  # progressive_count 0, 1, 2 (connection.rs:385-387)
        ref = run_oracle(scene, (160, 96), sky, rnd, jit[k], pc=k, history=history)
        assert_frame_equal(outs[k], ref)
        history = ref.rgba8


def test_progressive_accumulation_float(device):
    scene = cases.headline_scene(3)
    sky, rnd = cases.inputs(256, 128)
    jit = [(0.1, -0.2), (-0.33, 0.27), (0.45, 0.05)]
    outs, _ = run_gpu(device, scene, (160, 96), sky, rnd, jit, frames=3, accum=N.RT_ACCUM_FLOAT32)
    hist = None
    for k in range(3):
        ref = run_oracle(scene, (160, 96), sky, rnd, jit[k], pc=k, history_f32=hist, accum=1)
        assert_frame_equal(outs[k], ref)
        hist = ref.rgbf32


@pytest.mark.parametrize("variant", [2, 3])
def test_stress_1024_spheres(device, variant):
    """BASELINE config 3 scene (1024 spheres + plane; beyond the reference's MAX_OBJECTS)."""
    scene = cases.stress_scene(1024, 8)
    sky, rnd = cases.inputs(256, 128)
    outs, st = run_gpu(device, scene, (160, 90), sky, rnd, (0.2, 0.3), max_objects="live", variant=variant)
    ref = run_oracle(scene, (160, 90), sky, rnd, (0.2, 0.3), max_objects="live")
    assert st.sphere_count == 1024 and st.plane_count == 1
    assert_frame_equal(outs[0], ref)


def test_ragged_size_and_partition(device):
    """Sizes that are not multiples of the 8x8 wave tile, and a 3-way strip partition whose
    pieces reassemble to the single-context image bit for bit."""
    scene = cases.headline_scene(4)
    sky, rnd = cases.inputs(256, 128)
    size = (131, 77)
    full, _ = run_gpu(device, scene, size, sky, rnd, (0.3, -0.1))
    ref = run_oracle(scene, size, sky, rnd, (0.3, -0.1))
    assert_frame_equal(full[0], ref)
    from rsraytracingv3_amd.partition import assemble_rows
    world, strip = 3, 8
    pieces = []
    for rank in range(world):
        device.set_partition(rank, world, strip)
        target = RenderTarget.new(device, size)
        rt = RandomTexture(device)
        rt.write(rnd)
        conn = Connection(scene, device, sky, rt)
        render_scene(device, conn, target, scene, (0.3, -0.1))
        device.sync()
        pieces.append(target.read_rgbf32())
    device.set_partition(0, 1, 8)
    whole = assemble_rows(pieces, size[1], strip)
    assert np.array_equal(whole.view(np.uint32), ref.rgbf32.view(np.uint32))


@pytest.mark.parametrize("refill_min", [1, 7, 64])
def test_refill_threshold_does_not_change_results(device, refill_min):
    """The persistent kernel hands pixels to whichever lane is free; the image must not care."""
    scene = cases.headline_scene(8)
    sky, rnd = cases.inputs(256, 128)
    outs, st = run_gpu(device, scene, (203, 117), sky, rnd, (0.11, 0.37), refill_min=refill_min, variant=2)
    ref = run_oracle(scene, (203, 117), sky, rnd, (0.11, 0.37))
    assert_frame_equal(outs[0], ref)
    assert st.last_segments == int(ref.segments.sum())


def test_kernel_sqrt_is_ieee_for_every_float(device):
    """The sphere test's square root (v_sqrt_f32 + residual selection) is the correctly rounded
    IEEE result for all 2^31 non-negative floats (compared on the device with hipcc's sqrtf,
    which test_device_math_bitwise pins against the host's)."""
    import ctypes as C
    bad = C.c_uint64(123)
    device.call("rt_selftest_sqrt", C.byref(bad))
    assert bad.value == 0


def test_tile_parallel_renderer_single_rank(device):
    """TorchTileRenderer without a process group: caller-owned torch send buffers
    (rt_bind_outputs), the torch stream (rt_set_stream) and double buffering, against the oracle."""
    import torch
    from rsraytracingv3_amd.distributed import TorchTileRenderer
    scene = cases.headline_scene(5)
    sky, rnd = cases.inputs(256, 128)
    size = (150, 93)
    device.set_partition(0, 1, 8)
    device.set_accum_mode(N.RT_ACCUM_SRGB8)
    device.set_outputs(N.RT_OUT_RGBA8)
    device.set_variant(-1)
    RenderTarget.new(device, size)
    rt = RandomTexture(device)
    rt.write(rnd)
    conn = Connection(scene, device, sky, rt)
    conn.update_buffers(size, scene, (0.0, 0.0))
    tiles = TorchTileRenderer(device, size[0], size[1], strip_rows=8)
    try:
        jit = [(0.2, 0.1), (-0.4, 0.3), (0.05, -0.45)]
        history = None
        frames = []
        for k, j in enumerate(jit):   # three accumulating frames through alternating buffers
            out = tiles.render(conn, j, k)   # returns frame k-1
            if out is not None:
                frames.append(out.clone())
        frames.append(tiles.finish().clone())
        assert len(frames) == 3
        for k, j in enumerate(jit):
            got = frames[k].cpu().numpy().view(np.uint8).reshape(size[1], size[0], 4)
            ref = run_oracle(scene, size, sky, rnd, j, pc=k, history=history)
            assert np.array_equal(got, ref.rgba8), "frame %d" % k
            history = ref.rgba8
    finally:
        tiles.finish()
        device.call("rt_bind_outputs", None, None, None)
        private = torch.cuda.Stream()   # hand the context a stream that outlives this test
        device.call("rt_set_stream", private.cuda_stream)
        device._keepalive_stream = private


@pytest.mark.parametrize("variant", [0, 1, 3])
def test_primary_ray_culling_never_disagrees_with_the_full_scan(device, variant):
    """RT_OPT_CULL = 2 re-does every culled primary scan in full inside the kernel and counts
    disagreements; cameras inside the sphere field, far away, looking up, wide and narrow fov."""
    from rsraytracingv3_amd.scene import Camera
    sky, rnd = cases.inputs(128, 64)
    views = [((55., 65., 55.), (0.8, -0.8, 0.), 70.), ((34.4, 37.3, 105.5), (0.5, -0.3, 0.), 70.),
             ((3., 2., 4.), (0.05, 2.2, 0.), 90.), ((0., 400., 0.), (1.55, 0., 0.), 20.),
             ((10., 6., -30.), (-0.3, 3.0, 0.), 5.), ((0., 5., 0.), (-1.2, 0.7, 0.), 89.)]
    for scene in (cases.headline_scene(3), cases.stress_scene(300, 2)):
        for pos, rot, fov in views:
            scene.camera = Camera(pos, rot, fov)
            for jitter in ((0.0, 0.0), (0.495, -0.495), (-0.495, 0.495)):
                device.set_option(N.RT_OPT_CULL, 2)
                outs, st = run_gpu(device, scene, (203, 117), sky, rnd, jitter, max_objects="live", variant=variant)
                assert st.last_cull_mismatches == 0, (pos, rot, fov, jitter)
                device.set_option(N.RT_OPT_CULL, 0)
                plain, _ = run_gpu(device, scene, (203, 117), sky, rnd, jitter, max_objects="live", variant=variant)
                assert np.array_equal(outs[0][0].view(np.uint32), plain[0][0].view(np.uint32))
    device.set_option(N.RT_OPT_CULL, 1)


def test_jitter_beyond_half_a_pixel_renders_without_culling(device):
    scene = cases.headline_scene(3)
    sky, rnd = cases.inputs(128, 64)
    outs, _ = run_gpu(device, scene, (120, 64), sky, rnd, (3.25, -2.5))
    ref = run_oracle(scene, (120, 64), sky, rnd, (3.25, -2.5))
    assert_frame_equal(outs[0], ref)


def test_device_math_bitwise(device):
    """The device build of every specified math function equals the oracle's, bit for bit."""
    import ctypes as C
    rs = np.random.RandomState(3)
    n = 1 << 20

    def dev_unary(op, x):
        y = np.empty_like(x)
        device.call("rt_selftest_math_unary", op, x.ctypes.data_as(C.c_void_p), y.ctypes.data_as(C.c_void_p), x.size)
        return y

    def dev_binary(op, a, b):
        y = np.empty_like(a)
        device.call("rt_selftest_math_binary", op, a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                    y.ctypes.data_as(C.c_void_p), a.size)
        return y

    x = rs.uniform(-7, 7, n).astype(np.float32)
    for op, name in enumerate(["sin", "cos", "tan"]):
        assert np.array_equal(dev_unary(op, x).view(np.uint32), oracle.math_unary(name, x).view(np.uint32)), name
    x = rs.uniform(-1.01, 1.01, n).astype(np.float32)
    assert np.array_equal(dev_unary(3, x).view(np.uint32), oracle.math_unary("asin", x).view(np.uint32))
    x = np.exp(rs.uniform(-60, 60, n)).astype(np.float32)
    x[:4] = [0.0, -1.0, 1e-40, np.inf]
    assert np.array_equal(dev_unary(4, x).view(np.uint32), oracle.math_unary("log2", x).view(np.uint32))
    x = rs.uniform(-140, 140, n).astype(np.float32)
    assert np.array_equal(dev_unary(5, x).view(np.uint32), oracle.math_unary("exp2", x).view(np.uint32))
    x = np.exp(rs.uniform(-80, 80, n)).astype(np.float32)
    x[:3] = [0.0, 1e-42, 3e-39]
    assert np.array_equal(dev_unary(6, x).view(np.uint32), np.sqrt(x).view(np.uint32)), "device sqrt not IEEE"
    assert np.array_equal(dev_unary(7, x).view(np.uint32), np.sqrt(x).view(np.uint32)), "kernel sqrt_rn not IEEE"
    a = rs.normal(size=n).astype(np.float32)
    b = rs.normal(size=n).astype(np.float32)
    a[:4], b[:4] = [0, 0, 1, -1], [0, -1, 0, 0]
    assert np.array_equal(dev_binary(0, a, b).view(np.uint32), oracle.math_binary("atan2", a, b).view(np.uint32))
    a = rs.uniform(0, 1, n).astype(np.float32)
    b = rs.uniform(1e-3, 3, n).astype(np.float32)
    a[:2] = [0.0, 1e-40]
    assert np.array_equal(dev_binary(1, a, b).view(np.uint32), oracle.math_binary("pow", a, b).view(np.uint32))
    a = rs.normal(size=n).astype(np.float32)
    b = rs.normal(size=n).astype(np.float32)
    assert np.array_equal(dev_binary(2, a, b).view(np.uint32), (a / b).view(np.uint32)), "device divide not IEEE"
    # texel addressing (connection.rs:164-174 nearest + clamp-to-edge): clamp(floor(u * size), 0, size - 1), NaN -> 0
    u = rs.uniform(-0.5, 1.5, n).astype(np.float32)
    u[:8] = [np.nan, np.inf, -np.inf, 0.0, -0.0, 1.0, np.nextafter(np.float32(1), np.float32(0)), 1e30]
    size = rs.choice([1.0, 2.0, 600.0, 2048.0, 8192.0], n).astype(np.float32)
    with np.errstate(invalid="ignore"):
        t = np.floor(u * size)
        want = np.where(np.isnan(t), np.float32(0), np.minimum(np.maximum(t, np.float32(0)), size - np.float32(1))).astype(np.float32)
    assert np.array_equal(dev_binary(5, u, size), want), "texel index"
    # ... and the oracle's own addressing on directions that land on those edge cases (u = 0.5 + atan2/2pi, v = 0.5 + asin/pi)
    for d in ([0.0, 1.0, 0.0], [0.0, -1.0, 0.0], [0.0, 0.0, -1.0], [np.nan, 0.0, 1.0], [1e-30, 0.0, -1.0]):
        i, j = oracle.probe_hdri_texel(np.array(d, np.float32), 2048, 1024)
        assert 0 <= i < 2048 and 0 <= j < 1024


def _batch_vs_sequential(device, scene, size, sky, rnd, jitters, first_pc, accum, variant, partition=(0, 1, 8), max_objects="live"):
    """Renders len(jitters) frames once as separate launches and once as ONE rt_render_frames launch from the
    same starting history; returns the two (rgba8, rgbf32) states after the last frame plus the stats."""
    device.set_partition(*partition)
    device.set_accum_mode(accum)
    device.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32)
    device.set_variant(variant)
    target = RenderTarget.new(device, size)
    rt = RandomTexture(device)
    rt.write(rnd)
    conn = Connection(scene, device, sky, rt, max_objects=max_objects)
    conn.update_buffers(size, scene, (0.0, 0.0))
    states = []
    for mode in ("sequential", "batch"):
        # common starting history: two plain frames (counts 0, 1) so that first_pc > 0 mixes with real texels
        for k, j in enumerate([(0.11, 0.07), (-0.2, 0.3)]):
            conn.frame_data.jitter, conn.frame_data.progressive_count = j, k
            conn.draw()
        if mode == "sequential":
            for i, j in enumerate(jitters):
                conn.frame_data.jitter, conn.frame_data.progressive_count = (float(j[0]), float(j[1])), first_pc + i
                conn.draw()
        else:
            conn.draw_frames(jitters, first_pc)
        device.sync()
        states.append((target.read_rgba8().copy(), target.read_rgbf32().copy()))
    return states, device.stats()


@pytest.mark.parametrize("variant", [-1, 0, 1, 2, 3, 4])
@pytest.mark.parametrize("accum", [N.RT_ACCUM_SRGB8, N.RT_ACCUM_FLOAT32])
def test_batch_launch_equals_sequential_frames(device, variant, accum):
    """rt_render_frames: 5 frames in one launch == 5 rt_render_frame calls, texels and floats bit for bit."""
    scene = cases.headline_scene(6)
    sky, rnd = cases.inputs(256, 128)
    jit = [(0.1, -0.2), (-0.33, 0.27), (0.45, 0.05), (-0.49, -0.49), (0.0, 0.31)]
    (seq, bat), st = _batch_vs_sequential(device, scene, (203, 101), sky, rnd, jit, 2, accum, variant)
    assert st.last_batch == 5 and st.last_pixels == 5 * 203 * 101
    assert np.array_equal(seq[0], bat[0])
    assert np.array_equal(seq[1].view(np.uint32), bat[1].view(np.uint32))


@pytest.mark.parametrize("case", ["headline_live", "stress_300"])
def test_xcd_band_map_of_multi_frame_launches(device, case):
    """render_blocks deals the tiles of a multi-frame launch to the XCDs in vertical bands (block_tile_pos) when its grid is a multiple
    of 8 workgroups wide — 128 pixels: 8 two-wave workgroups, 16 one-wave ones with candidate lists — and rotates the bands every 32 tile
    rows and every frame: 42 tile rows (the last one ragged) x 3 frames in ONE launch == the same frames one launch each (plain map), bit for
    bit, and every pixel-frame rendered exactly once (the launch's pixel count)."""
    scene = cases.headline_scene(6) if case == "headline_live" else cases.stress_scene(300, 6)
    sky, rnd = cases.inputs(256, 128)
    jit = [(0.1, -0.2), (-0.33, 0.27), (0.45, 0.05)]
    (seq, bat), st = _batch_vs_sequential(device, scene, (128, 330), sky, rnd, jit, 2, N.RT_ACCUM_SRGB8, 3)
    assert st.last_variant == 3 and st.last_batch == 3 and st.last_pixels == 3 * 128 * 330
    assert np.array_equal(seq[0], bat[0])
    assert np.array_equal(seq[1].view(np.uint32), bat[1].view(np.uint32))


@pytest.mark.parametrize("refill_min,tile_chunk", [(1, 1), (1, 2), (9, 3), (64, 1)])
def test_streamed_shape_primary_pass_queue_bounce_pass(device, refill_min, tile_chunk):
    """Variant 4 (primary pass -> path queue -> resident waves refilled from the queue) in a multi-frame launch: the
    launch really takes that shape, and texels, floats and the segment / rough-hit totals equal the sequential frames'
    whatever the refill threshold and the chunk a wave takes off the queue."""
    scene = cases.headline_scene(24)
    sky, rnd = cases.inputs(256, 128)
    jit = [(0.1, -0.2), (-0.33, 0.27), (0.45, 0.05), (-0.49, -0.49), (0.0, 0.31), (0.2, 0.2)]
    device.set_option(N.RT_OPT_REFILL_MIN, refill_min)
    device.set_option(N.RT_OPT_TILE_CHUNK, tile_chunk)
    try:
        (seq, bat), st = _batch_vs_sequential(device, scene, (333, 187), sky, rnd, jit, 0, N.RT_ACCUM_FLOAT32, 4)
        assert st.last_variant == (4 if AB else 3) and st.last_batch == 6
        assert np.array_equal(seq[0], bat[0])
        assert np.array_equal(seq[1].view(np.uint32), bat[1].view(np.uint32))
        batch_segments, batch_rough = st.last_segments, st.last_rough_hits
        device.set_variant(3)
        (seq3, bat3), st3 = _batch_vs_sequential(device, scene, (333, 187), sky, rnd, jit, 0, N.RT_ACCUM_FLOAT32, 3)
        assert st3.last_variant == 3
        assert (st3.last_segments, st3.last_rough_hits) == (batch_segments, batch_rough)
        assert np.array_equal(bat3[1].view(np.uint32), bat[1].view(np.uint32))
    finally:
        device.set_option(N.RT_OPT_REFILL_MIN, 1)
        device.set_option(N.RT_OPT_TILE_CHUNK, 2)


@pytest.mark.parametrize("case", ["headline_live", "reference_80_slots", "mirrors", "rank_1_of_3"])
def test_tail_spill_hands_the_last_paths_to_full_waves(device, case):
    """RT_OPT_TAIL_SPILL: a workgroup of the re-packing shape with at most T paths left appends them, with their pending
    hits, to the path queue, and tail_stream shades and continues them in refilled waves.  Texels, floats and the segment /
    rough-hit totals of a multi-frame launch equal the sequential frames' for every threshold (4 = late bounces only ...
    128 = everything after the primary scan), refill threshold and chunk size."""
    from rsraytracingv3_amd.scene import Camera
    size, slots, partition = (333, 187), "live", (0, 1, 8)
    scene = cases.headline_scene(24)
    if case == "reference_80_slots":   # 79 spheres: two-word candidate sets, the 6-wave instance
        slots = None
    elif case == "mirrors":            # long paths between smooth spheres: the tail runs most of the bounces
        for o in scene.objects:
            o.material.roughness = 0.0
        scene.camera = Camera((30.0, 6.0, 30.0), (0.1, -2.3, 0.0), 75.0)
    elif case == "rank_1_of_3":
        size, partition = (250, 203), (1, 3, 8)
    sky, rnd = cases.inputs(256, 128)
    jit = [(0.1, -0.2), (-0.33, 0.27), (0.45, 0.05), (-0.49, -0.49), (0.0, 0.31), (0.2, 0.2)]
    try:
        device.set_option(N.RT_OPT_FLAGS, 4)  # launches of every size spill (by default only those of 100 000 tiles or more)
        device.set_option(N.RT_OPT_TAIL_SPILL, 0)
        device.set_option(N.RT_OPT_TAIL_SPILL_SINGLE, 0)
        (seq, bat0), st0 = _batch_vs_sequential(device, scene, size, sky, rnd, jit, 0, N.RT_ACCUM_FLOAT32, 3, partition, slots)
        assert st0.last_variant == 3 and st0.last_stage < 100 and st0.dirmask_cells > 0
        assert np.array_equal(seq[0], bat0[0]) and np.array_equal(seq[1].view(np.uint32), bat0[1].view(np.uint32))
        for t, refill_min, chunk in ((4, 1, 1), (24, 16, 2), (64, 9, 3), (128, 64, 1), (32, 1, 2)):
            device.set_option(N.RT_OPT_TAIL_SPILL, t)
            device.set_option(N.RT_OPT_TAIL_SPILL_SINGLE, t)
            device.set_option(N.RT_OPT_REFILL_MIN, refill_min)
            device.set_option(N.RT_OPT_TILE_CHUNK, chunk)
            (seq_t, bat), st = _batch_vs_sequential(device, scene, size, sky, rnd, jit, 0, N.RT_ACCUM_FLOAT32, 3, partition, slots)
            assert st.last_variant == 3 and st.last_stage >= 100, "the launch did not take the spilling shape"
            assert (st.last_segments, st.last_rough_hits) == (st0.last_segments, st0.last_rough_hits)
            assert np.array_equal(seq[0], bat[0]), "threshold %d: texels differ" % t
            assert np.array_equal(seq[1].view(np.uint32), bat[1].view(np.uint32)), "threshold %d: floats differ" % t
            # the sequential frames of that run were single-frame launches in flight, which spill as well
            assert np.array_equal(seq[0], seq_t[0]) and np.array_equal(seq[1].view(np.uint32), seq_t[1].view(np.uint32))
            tail_ms = np.zeros(1, np.float32)
            device.call("rt_get_tail_history", 1, tail_ms.ctypes.data_as(C.c_void_p))
            assert 0.0 < tail_ms[0] < st.last_kernel_ms, "tail_stream's time is part of the launch's kernel time"
        # by default a launch this small does not spill
        device.set_option(N.RT_OPT_FLAGS, 0)
        device.set_option(N.RT_OPT_TAIL_SPILL, 32)
        (_, bat), st = _batch_vs_sequential(device, scene, size, sky, rnd, jit, 0, N.RT_ACCUM_FLOAT32, 3, partition, slots)
        assert st.last_stage < 100 and np.array_equal(seq[0], bat[0])
        # a launch with per-frame diagnostic outputs is not sample-only: no spill
        outs, st1 = run_gpu(device, scene, (96, 64), sky, rnd, (0.0, 0.0), variant=3, max_objects=slots)
        assert st1.last_stage < 100
        assert_frame_equal(outs[0], run_oracle(scene, (96, 64), sky, rnd, (0.0, 0.0), max_objects=slots))
    finally:
        device.set_option(N.RT_OPT_FLAGS, 0)
        device.set_option(N.RT_OPT_TAIL_SPILL, N.RT_TAIL_SPILL_DEFAULT)
        device.set_option(N.RT_OPT_TAIL_SPILL_SINGLE, N.RT_TAIL_SPILL_SINGLE_DEFAULT)
        device.set_option(N.RT_OPT_REFILL_MIN, 1)
        device.set_option(N.RT_OPT_TILE_CHUNK, 2)
        device.set_partition(0, 1, 8)


def test_streamed_shape_under_a_partition_and_its_fall_back(device):
    """Variant 4 on rank 1's strips of a 3-way partition (the strip mapping of path_begin, the queue sized for the
    rank's rows) equals the sequential frames; a launch that is not sample-only (one frame at a time, nothing in flight)
    asked for variant 4 takes the re-packing shape instead."""
    scene = cases.headline_scene(12)
    sky, rnd = cases.inputs(256, 128)
    jit = [(0.1, -0.2), (-0.33, 0.27), (0.45, 0.05), (-0.49, -0.49)]
    (seq, bat), st = _batch_vs_sequential(device, scene, (250, 203), sky, rnd, jit, 0, N.RT_ACCUM_SRGB8, 4, partition=(1, 3, 8))
    assert st.last_variant == (4 if AB else 3) and st.last_batch == 4
    assert np.array_equal(seq[0], bat[0]) and np.array_equal(seq[1].view(np.uint32), bat[1].view(np.uint32))
    device.set_option(N.RT_OPT_INFLIGHT, 1)
    try:
        outs, st1 = run_gpu(device, scene, (96, 64), sky, rnd, (0.0, 0.0), variant=4)
        assert st1.last_variant == 3
        assert_frame_equal(outs[0], run_oracle(scene, (96, 64), sky, rnd, (0.0, 0.0)))
    finally:
        device.set_option(N.RT_OPT_INFLIGHT, 0)
        device.set_partition(0, 1, 8)


def test_batch_launch_restarting_accumulation_and_oracle(device):
    """A batch whose first frame has progressive_count 0 (history ignored) against the oracle's frame chain,
    under a 3-way strip partition (rank 1's rows)."""
    scene = cases.headline_scene(4)
    sky, rnd = cases.inputs(256, 128)
    size = (96, 80)
    jit = [(0.25, 0.25), (-0.25, 0.1), (0.4, -0.4)]
    (seq, bat), _ = _batch_vs_sequential(device, scene, size, sky, rnd, jit, 0, N.RT_ACCUM_SRGB8, -1, partition=(1, 3, 8))
    assert np.array_equal(seq[0], bat[0]) and np.array_equal(seq[1].view(np.uint32), bat[1].view(np.uint32))
    history = None
    for k in range(3):
        ref = oracle.render(scene.live_object_bytes(), len(scene.objects), scene.config_bytes(*size), jit[k], k, sky, rnd,
                            history_rgba8=history)
        history = ref.rgba8
    from rsraytracingv3_amd.partition import local_row_indices
    rows = local_row_indices(size[1], 1, 3, 8)
    assert np.array_equal(bat[0], ref.rgba8[rows])
    assert np.array_equal(bat[1].view(np.uint32), ref.rgbf32[rows].view(np.uint32))


def test_batch_launch_limits(device):
    """64 frames is the most one launch takes; the single-frame debug outputs must be off."""
    scene = cases.config1_scene()
    sky, rnd = cases.inputs(64, 32)
    device.set_partition(0, 1, 8)
    device.set_outputs(N.RT_OUT_RGBA8)
    target = RenderTarget.new(device, (64, 48))
    rt = RandomTexture(device)
    rt.write(rnd)
    conn = Connection(scene, device, sky, rt)
    conn.update_buffers((64, 48), scene, (0.0, 0.0))
    conn.draw_frames([(0.0, 0.0)] * 64, 0)
    device.sync()
    assert device.stats().last_batch == 64
    with pytest.raises(N.RtError):
        conn.draw_frames([(0.0, 0.0)] * 65, 0)
    device.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_SEGMENTS)
    RenderTarget.new(device, (64, 48))
    with pytest.raises(N.RtError):
        conn.draw_frames([(0.0, 0.0)] * 2, 0)
    conn.draw_frames([(0.1, 0.1)], 0)  # a batch of one is a plain frame, debug outputs allowed
    device.sync()
    assert target.read_segments().shape == (48, 64)


@pytest.mark.parametrize("accum", [N.RT_ACCUM_SRGB8, N.RT_ACCUM_FLOAT32])
def test_two_frames_in_flight_equal_one_at_a_time_and_the_oracle(device, accum):
    """RT_OPT_INFLIGHT (default 4; here 4, 2 and 1): single-frame launches (the reference's one draw per loop turn,
    app.rs:208-254) alternate between two internal streams so that frame k + 1 starts while frame k drains; the
    history is folded in frame order.  12 frames without a sync in between, the random texture replaced on the
    device every 4 frames: same texels and floats as one launch at a time, and as the oracle's frame chain."""
    scene = cases.headline_scene(5)
    size = (200, 120)
    sky, _ = cases.inputs(256, 128)
    jit = synth.jitter_sequence(9, 12)
    results = {}
    for inflight in (0, 5, 4, 2, 1):  # 0 = the library's rule: four, five for short launches (paths of one or two segments, or fewer tiles than a 1080p frame: this one)
        device.set_partition(0, 1, 8)
        device.set_accum_mode(accum)
        device.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32)
        device.set_variant(-1)
        device.set_option(N.RT_OPT_INFLIGHT, inflight)
        target = RenderTarget.new(device, size)
        rt = RandomTexture(device, seed=70)
        rt.generate_on_device()
        conn = Connection(scene, device, sky, rt, max_objects="live")
        conn.update_buffers(size, scene, (0.0, 0.0))
        rt.seed = 70
        for k in range(12):
            if k % 4 == 0:
                rt.generate_on_device()
            conn.frame_data.jitter = (float(jit[k, 0]), float(jit[k, 1]))
            conn.frame_data.progressive_count = k
            conn.draw()
        device.sync()
        assert device.stats().last_launches_in_flight == (inflight or 5)
        results[inflight] = (target.read_rgbf32(), target.read_rgba8())
    device.set_option(N.RT_OPT_INFLIGHT, 0)
    short = cases.headline_scene(2)
    conn = Connection(short, device, sky, rt, max_objects="live")
    conn.update_buffers(size, short, (0.0, 0.0))
    conn.draw()
    device.sync()
    assert device.stats().last_launches_in_flight == 5
    assert np.array_equal(results[0][0].view(np.uint32), results[1][0].view(np.uint32)) and np.array_equal(results[0][1], results[1][1])
    assert np.array_equal(results[5][0].view(np.uint32), results[1][0].view(np.uint32)) and np.array_equal(results[5][1], results[1][1])
    assert np.array_equal(results[4][0].view(np.uint32), results[1][0].view(np.uint32)) and np.array_equal(results[4][1], results[1][1])
    assert np.array_equal(results[2][0].view(np.uint32), results[1][0].view(np.uint32))
    assert np.array_equal(results[2][1], results[1][1])
    hist8 = histf = None
    for k in range(12):
        ref = run_oracle(scene, size, sky, synth.random_texture(70 + k // 4), jit[k], max_objects="live", pc=k,
                         history=hist8, history_f32=histf, accum=1 if accum == N.RT_ACCUM_FLOAT32 else 0)
        hist8, histf = (None, ref.rgbf32) if accum == N.RT_ACCUM_FLOAT32 else (ref.rgba8, None)
    assert np.array_equal(results[2][0].view(np.uint32), ref.rgbf32.view(np.uint32))
    assert np.array_equal(results[2][1], ref.rgba8)


@pytest.mark.parametrize("accum", [N.RT_ACCUM_SRGB8, N.RT_ACCUM_FLOAT32])
def test_intermediate_frames_of_a_batch_launch(device, accum):
    """rt_keep_batch_frames: the reference presents every frame (app.rs:240-253); a multi-frame launch can keep the
    stored texels after every k-th frame.  Kept frame i == the target after frames 0..i rendered one per launch."""
    import ctypes as C
    scene = cases.headline_scene(4)
    size = (150, 93)
    sky, rnd = cases.inputs(256, 128)
    jit = synth.jitter_sequence(11, 6)

    def setup():
        device.set_partition(0, 1, 8)
        device.set_accum_mode(accum)
        device.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32)
        device.set_variant(-1)
        target = RenderTarget.new(device, size)
        rt = RandomTexture(device)
        rt.write(rnd)
        conn = Connection(scene, device, sky, rt, max_objects="live")
        conn.update_buffers(size, scene, (0.0, 0.0))
        return target, conn

    target, conn = setup()
    sequential = []
    for k in range(6):
        conn.frame_data.jitter = (float(jit[k, 0]), float(jit[k, 1]))
        conn.frame_data.progressive_count = k
        conn.draw()
        device.sync()
        sequential.append(target.read_rgba8())
    try:
        for every in (1, 2, 3):
            target, conn = setup()
            device.call("rt_keep_batch_frames", every)
            conn.draw_frames(jit, 0)
            device.sync()
            assert np.array_equal(target.read_rgba8(), sequential[5])
            for i in range(6):
                out = np.empty((size[1], size[0], 4), np.uint8)
                rc = device._lib.rt_read_batch_frame(device.handle, i, out.ctypes.data_as(C.c_void_p), out.nbytes)
                if (i + 1) % every == 0:
                    assert rc == 0 and np.array_equal(out, sequential[i]), "every %d, frame %d" % (every, i)
                else:
                    assert rc == -1  # RT_ERR_INVALID_ARGUMENT: that frame was not kept
    finally:
        device.call("rt_keep_batch_frames", 0)


@pytest.mark.parametrize("accum", [N.RT_ACCUM_SRGB8, N.RT_ACCUM_FLOAT32])
def test_multi_frame_launches_in_flight_equal_one_at_a_time(device, accum):
    """RT_OPT_BATCH_INFLIGHT (default 0 = the library's rule, four for launches this small; here also 3, 2 and 1) with
    RT_OPT_INFLIGHT 4 / 4 / 2 / 1: a mixed sequence of multi-frame
    launches and single frames, queued without a sync, the random texture replaced on the device in between — launches
    overlap on the internal streams, the history is folded strictly in frame order, so texels and floats equal those of
    one launch at a time.  rt_get_launch_span: the device time of overlapping launches is less than the sum of their
    kernel times and at least the longest of them."""
    scene = cases.headline_scene(7)
    size = (264, 136)
    sky, _ = cases.inputs(256, 128)
    sequence = [5, 1, 3, 1, 1, 2, 8, 1, 4]  # frames per launch
    total = sum(sequence)
    jit = synth.jitter_sequence(11, total)
    results = {}
    try:
        for lanes_batch, lanes_single in ((0, 4), (3, 4), (2, 2), (1, 1)):
            device.set_partition(0, 1, 8)
            device.set_accum_mode(accum)
            device.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32)
            device.set_variant(-1)
            device.set_option(N.RT_OPT_BATCH_INFLIGHT, lanes_batch)
            device.set_option(N.RT_OPT_INFLIGHT, lanes_single)
            target = RenderTarget.new(device, size)
            rt = RandomTexture(device, seed=90)
            rt.generate_on_device()
            conn = Connection(scene, device, sky, rt, max_objects="live")
            conn.update_buffers(size, scene, (0.0, 0.0))
            rt.seed = 90
            k = 0
            for n, frames in enumerate(sequence):
                if n % 3 == 0:
                    rt.generate_on_device()
                if frames == 1:
                    conn.frame_data.jitter = (float(jit[k, 0]), float(jit[k, 1]))
                    conn.frame_data.progressive_count = k
                    conn.draw()
                else:
                    conn.draw_frames([(float(x), float(y)) for x, y in jit[k:k + frames]], k)
                k += frames
            device.sync()
            assert device.stats().last_launches_in_flight == (lanes_batch or 4)  # the last launch carried 4 frames
            n_launches = len(sequence)
            kms = np.zeros(n_launches, np.float32)
            device.call("rt_get_frame_history", n_launches, None, None, kms.ctypes.data_as(C.c_void_p))
            span = C.c_float(0.0)
            device.call("rt_get_launch_span", n_launches, C.byref(span))
            assert span.value >= float(kms.max()) * 0.999
            if lanes_batch == 1 and lanes_single == 1:
                assert span.value >= float(kms.sum()) * 0.999  # one at a time: the kernels do not overlap
            results[lanes_batch] = (target.read_rgbf32(), target.read_rgba8())
    finally:
        device.set_option(N.RT_OPT_BATCH_INFLIGHT, 1)
        device.set_option(N.RT_OPT_INFLIGHT, 0)
    for lanes in (0, 3, 2):
        assert np.array_equal(results[lanes][0].view(np.uint32), results[1][0].view(np.uint32))
        assert np.array_equal(results[lanes][1], results[1][1])
    with pytest.raises(N.RtError):
        device.set_option(N.RT_OPT_BATCH_INFLIGHT, 9)


def test_branch_free_root_gives_the_exact_hit_key(device):
    """The scans compute t = -dn - sqrt(q) with a root that is only correctly rounded for q >= 2^-96 (sqrt_for_hit,
    rt_kernels.hip); below that its value is some r in [0, 2^-47], which must give the same hit key as the IEEE root —
    bit for bit when the key can be accepted (below the initial distance's key), "never accepted" on both sides
    otherwise.  Swept over dn of both signs and every magnitude against q from denormals to 1e12, with the binade
    edges of the argument (2^-96 for q, 2^-22 for -dn, EPSILON for t) put in by hand."""
    rs = np.random.RandomState(11)
    n = 1 << 20

    def keys(op, dn, q):
        y = np.empty_like(dn)
        device.call("rt_selftest_math_binary", op, dn.ctypes.data_as(C.c_void_p), q.ctypes.data_as(C.c_void_p),
                    y.ctypes.data_as(C.c_void_p), dn.size)
        return y.view(np.uint32)

    no_hit = np.uint32(0x49742400 - 0x358637bd)  # key of the initial hit.distance 1e6 (ray.hlsl:98)
    mag = np.exp(rs.uniform(np.log(1e-45), np.log(1e7), n)).astype(np.float32)
    dn = (-mag * np.where(rs.uniform(size=n) < 0.85, 1.0, -1.0)).astype(np.float32)
    q = np.exp(rs.uniform(np.log(1e-46), np.log(1e12), n)).astype(np.float32)
    tiny = rs.uniform(size=n) < 0.5
    q[tiny] = np.exp(rs.uniform(np.log(1e-46), np.log(2.0 ** -94), int(tiny.sum()))).astype(np.float32)
    edge_q = np.array([0.0, 1e-45, 2.0 ** -126, 2.0 ** -97, np.nextafter(np.float32(2.0 ** -96), np.float32(0)), 2.0 ** -96,
                       np.nextafter(np.float32(2.0 ** -96), np.float32(1)), 1e-30, 1.0], np.float32)
    edge_dn = -np.array([0.0, 1e-45, 1e-7, np.nextafter(np.float32(2.0 ** -22), np.float32(0)), 2.0 ** -22,
                         np.nextafter(np.float32(2.0 ** -22), np.float32(1)), 9.99999e-7, 1e-6, 1.0000001e-6, 3e-6, 1.0, 4.0,
                         999999.9, 1e6], np.float32)
    gq, gd = np.meshgrid(edge_q, edge_dn)
    k = gq.size
    q[:k], dn[:k] = gq.ravel(), gd.ravel()
    fast, exact = keys(3, dn, q), keys(4, dn, q)
    with np.errstate(invalid="ignore"):
        ref = ((-dn - np.sqrt(q)).astype(np.float32).view(np.uint32) - np.uint32(0x358637bd))
    assert np.array_equal(exact, ref), "device IEEE root key differs from numpy"
    same = (fast == exact) | ((fast >= no_hit) & (exact >= no_hit))
    assert same.all(), "%d of %d keys differ, e.g. dn=%r q=%r" % ((~same).sum(), n, dn[~same][:3], q[~same][:3])
    assert (fast[q >= np.float32(2.0 ** -96)] == exact[q >= np.float32(2.0 ** -96)]).all()
    assert int(((exact < no_hit) & (q < np.float32(2.0 ** -96))).sum()) > 1000, "the sweep must reach acceptable keys with tiny q"


@pytest.mark.parametrize("variant", [0, 3])
def test_degenerate_spheres_around_ray_origins(device, variant):
    """Discriminants far below 2^-96, where the scans' root is only bounded, not correctly rounded (sqrt_for_hit): spheres
    of radius 1e-17 .. 1e-9 centred exactly at the camera, a hair in front of it along the axes and the view direction,
    and at the points where a big sphere's bounce rays start — against the oracle's IEEE roots, with the hit indices
    compared per segment."""
    from rsraytracingv3_amd.scene import Camera, Geometry, Material, Object, Scene
    cam = (1.5, 2.0, -6.0)
    objs = [Object("big", Material((0.9, 0.8, 0.7), (0, 0, 0), 0.0, 0.5, 0.0), Geometry.sphere((1.5, 2.0, 4.0), 3.0)),
            Object("rough", Material((0.5, 0.9, 0.6), (1, 1, 0), 2.0, 0.5, 0.6), Geometry.sphere((-4.0, 1.0, 6.0), 2.5))]
    k = 0
    for radius in (1e-17, 1e-15, 1e-13, 1e-11, 1e-9):
        for off in ((0, 0, 0), (1e-12, 0, 0), (0, 0, 2.0 ** -10), (0, 3e-8, 1e-7), (0, 0, 1e-3), (2.0 ** -12, 0, 2.0 ** -12)):
            c = tuple(float(np.float32(a) + np.float32(b)) for a, b in zip(cam, off))
            objs.append(Object("dust%d" % k, Material((0.3, 0.3, 0.9), (0, 1, 1), 1.0, 0.5, 0.3), Geometry.sphere(c, radius)))
            k += 1
    # dust on the big sphere's surface, where mirror bounces start 0.001 above it
    for a in np.linspace(-0.6, 0.6, 9):
        p = (1.5 + 3.0 * np.sin(a), 2.0, 4.0 - 3.0 * np.cos(a))
        n = (np.sin(a), 0.0, -np.cos(a))
        c = tuple(float(np.float32(pi + 0.001 * ni)) for pi, ni in zip(p, n))
        objs.append(Object("dust%d" % k, Material((0.9, 0.2, 0.2), (1, 0, 0), 3.0, 0.5, 0.0), Geometry.sphere(c, 1e-12)))
        k += 1
    objs.append(Object("ground", Material((0.5, 0.5, 0.5), (0, 0, 0), 0.0, 0.5, 0.5), Geometry.plane((0, -1.0, 0), (0, 1, 0), 1e5)))
    scene = Scene(objects=objs, camera=Camera(cam, (0.0, 0.0, 0.0), 60.0), reflection_limit=6)
    sky, rnd = cases.inputs(128, 64)
    for size, jitter in (((65, 65), (0.0, 0.0)), ((128, 96), (0.25, -0.125))):  # 65x65, no jitter: a pixel centre on the view axis
        outs, st = run_gpu(device, scene, size, sky, rnd, jitter, max_objects="live", variant=variant)
        ref = run_oracle(scene, size, sky, rnd, jitter, max_objects="live")
        assert_frame_equal(outs[0], ref)
        assert st.last_segments == int(ref.segments.sum())


@pytest.mark.parametrize("variant", [0, 3])
@pytest.mark.parametrize("kind", ["bit_sets", "lists"])
def test_a_planet_sized_ground_sphere_and_a_sky_dome_are_outliers(device, variant, kind):
    """The other shape of an outlier: not far away but huge — a ground SPHERE of radius 5 000 under the scene (the paths bounce off it, so
    rays start on an outlier's surface in the middle of the table) and a dome of radius 3 000 around everything.  |c - median| + r puts
    both beyond the limit: they do not size the tables and sit in every set.  Frame, segments and hit indices are the oracle's;
    RT_OPT_CULL = 2 counts no disagreement."""
    from rsraytracingv3_amd.scene import Geometry, Material, Object
    scene = cases.headline_scene(5) if kind == "bit_sets" else cases.stress_scene(300, 5)
    scene.objects = [o for o in scene.objects if o.geometry.kind == 0]  # no plane: the planet is the ground
    scene.objects.insert(9, Object("planet", Material((0.6, 0.7, 0.5), (0, 0, 0), 0.0, 0.5, 0.3), Geometry.sphere((3.0, -5000.0, -2.0), 5000.0)))
    scene.objects.insert(30, Object("dome", Material((0.8, 0.8, 1.0), (0.2, 0.3, 1.0), 1.0, 0.5, 1.0), Geometry.sphere((0.0, 0.0, 0.0), 3000.0)))
    sky, rnd = cases.inputs(128, 64)
    size, jitter = (192, 108), (0.15, -0.25)
    ref = run_oracle(scene, size, sky, rnd, jitter, max_objects="live")
    device.set_option(N.RT_OPT_CULL, 2)
    try:
        outs, st = run_gpu(device, scene, size, sky, rnd, jitter, max_objects="live", variant=variant)
    finally:
        device.set_option(N.RT_OPT_CULL, 1)
    assert_frame_equal(outs[0], ref)
    assert st.last_cull_mismatches == 0 and st.dirmask_cells > 0 and (st.dirmask_lists > 0) == (kind == "lists")
    assert (ref.hits == 9).any()  # the planet is hit (the dome, seen from inside, never is: ray.hlsl:53-68 takes the near root only)


@pytest.mark.parametrize("variant", [0, 3])
@pytest.mark.parametrize("kind", ["bit_sets", "lists"])
def test_spheres_far_from_the_others_do_not_size_the_tables(device, variant, kind):
    """rt_dirmask.h "Outliers": the headline scene with three far spheres — a big mirror in view that paths bounce off and back into
    the scene, a tiny one, one overhead —: they do not size the tables and are members of every candidate set instead.  Frames, segment
    counts and hit indices are the oracle's, from two cameras (one of them next to the far mirror: rays that start outside the near box
    scan linearly), RT_OPT_CULL = 2 counts no disagreement with the linear scan, and with RT_OPT_FLAGS bit 5 (the outliers stay in the
    table) the frames are the same."""
    from rsraytracingv3_amd.scene import Camera, Geometry, Material, Object
    scene = cases.headline_scene(6) if kind == "bit_sets" else cases.stress_scene(300, 6)  # (the same for the candidate lists)
    mirror = Material((0.9, 0.9, 0.9), (0, 0, 0), 0.0, 0.5, 0.0)
    rough = Material((0.9, 0.4, 0.3), (1.0, 1.0, 1.0), 0.5, 0.5, 0.7)
    f = 1.0 if kind == "bit_sets" else 3.0
    far = [(Geometry.sphere((-420.0 * f, 60.0 * f, -380.0 * f), 90.0 * f), mirror), (Geometry.sphere((-900.0 * f, 500.0 * f, 100.0 * f), 1.0), rough),
           (Geometry.sphere((30.0 * f, 900.0 * f, 30.0 * f), 25.0 * f), rough)]
    for k, (g, m) in enumerate(far):
        scene.objects.insert(10 * k + 3, Object("far%d" % k, m, g))
    sky, rnd = cases.inputs(128, 64)
    size, jitter = (192, 108), (-0.1, 0.35)
    for cam in (scene.camera, Camera((-300.0 * f, 80.0 * f, -270.0 * f), (0.2, 2.4, 0.0), 70.0)):
        scene.camera = cam
        ref = run_oracle(scene, size, sky, rnd, jitter, max_objects="live")
        for flags in (0, 32):
            device.set_option(N.RT_OPT_FLAGS, flags)
            device.set_option(N.RT_OPT_CULL, 2)
            try:
                outs, st = run_gpu(device, scene, size, sky, rnd, jitter, max_objects="live", variant=variant)
            finally:
                device.set_option(N.RT_OPT_FLAGS, 0)
                device.set_option(N.RT_OPT_CULL, 1)
            assert_frame_equal(outs[0], ref)
            assert st.last_cull_mismatches == 0 and st.dirmask_cells > 0 and (st.dirmask_lists > 0) == (kind == "lists")
            assert st.last_segments == int(ref.segments.sum())


@pytest.mark.parametrize("variant", [0, 3])
def test_repeated_sphere_records_are_dropped_and_the_first_one_wins(device, variant):
    """rt_set_scene leaves sphere records that repeat an earlier one bit for bit out of the scene image (the zero-filled tail of the
    reference's 80 slots: 15 -> 1).  The scan's strict '<' in ascending object order (ray.hlsl:98-106) lets the FIRST occurrence win
    whatever the materials: copies of three spheres of the headline scene with other materials, spread over the buffer, plus the zero
    tail — frame, segment counts and hit indices are the oracle's (it scans all 80), with the copies dropped and (RT_OPT_FLAGS bit 3)
    with them kept."""
    import copy
    from rsraytracingv3_amd.scene import Material, Object
    scene = cases.headline_scene(6)
    scene.objects = scene.objects[:40] + [scene.objects[-1]]  # 40 spheres + the ground plane
    spheres = [o for o in scene.objects if o.geometry.kind == 0]
    for k, src in enumerate((spheres[3], spheres[17], spheres[3], spheres[30])):
        twin = Object("twin%d" % k, Material((0.1, 0.9, 0.2), (1.0, 0.0, 1.0), 5.0, 0.5, 0.9 if k % 2 else 0.0), copy.deepcopy(src.geometry))
        scene.objects.insert(7 * k + 5, twin) if k < 2 else scene.objects.append(twin)
    n_wire = sum(1 for o in scene.objects if o.geometry.kind == 0)
    sky, rnd = cases.inputs(128, 64)
    size, jitter = (160, 96), (0.2, -0.3)
    ref = run_oracle(scene, size, sky, rnd, jitter)  # the reference's 80 slots
    for keep in (0, 8):
        device.set_option(N.RT_OPT_FLAGS, keep)
        try:
            outs, st = run_gpu(device, scene, size, sky, rnd, jitter, variant=variant)
        finally:
            device.set_option(N.RT_OPT_FLAGS, 0)
        assert_frame_equal(outs[0], ref)
        zero_tail = 80 - len(scene.objects)
        assert st.sphere_count == n_wire + zero_tail
        # dropped: the two copies behind their originals count; a copy placed BEFORE its original makes the original the repeat
        assert st.spheres_scanned == (st.sphere_count if keep else n_wire - 4 + 1)


@pytest.mark.parametrize("variant", [0, 2, 3])
@pytest.mark.parametrize("case", ["headline_live", "reference_80_slots", "stress_100", "stress_128", "inside_the_scene"])
def test_direction_mask_table_never_changes_a_frame(device, variant, case):
    """rt_dirmask.h: secondary rays of scenes with 26..128 spheres test only the spheres of their table entry.  The frame
    must be the oracle's (= the linear scan's) bit for bit, with the table at several resolutions, one and two 64-bit
    words per entry (the reference's 80 slots hold 79 spheres), and with RT_OPT_CULL = 2 the kernel re-scans every
    lookup linearly and counts disagreements: zero."""
    from rsraytracingv3_amd.scene import Camera
    size, jitter, slots = (200, 112), (0.3, -0.2), "live"
    if case == "headline_live":
        scene = cases.headline_scene(8)
    elif case == "reference_80_slots":
        scene, slots = cases.headline_scene(8), None
    elif case == "stress_100":
        scene = cases.stress_scene(100, 6)
        scene.camera = Camera((60.0, 40.0, 60.0), (0.5, -0.8, 0.0), 70.0)
    elif case == "stress_128":
        scene = cases.stress_scene(128, 6)
        scene.camera = Camera((60.0, 40.0, 60.0), (0.5, -0.8, 0.0), 70.0)
    else:  # camera between the spheres, looking along the ground: long shallow rays, origins outside the spheres' box
        scene = cases.headline_scene(8)
        scene.camera = Camera((70.0, 4.0, 10.0), (0.02, -1.6, 0.0), 80.0)
    sky, rnd = cases.inputs(64, 32)
    ref = run_oracle(scene, size, sky, rnd, jitter, max_objects=slots)
    try:
        for k, cells in ((12, 4096), (3, 64), (16, 512)):
            device.set_option(N.RT_OPT_DIRMASK, 1)
            device.set_option(N.RT_OPT_DIRMASK_K, k)
            device.set_option(N.RT_OPT_DIRMASK_CELLS, cells)
            device.set_option(N.RT_OPT_CULL, 2)
            outs, st = run_gpu(device, scene, size, sky, rnd, jitter, max_objects=slots, variant=variant)
            assert st.dirmask_cells > 0 and st.dirmask_bins == 6 * k * k, "the scene got no table"
            assert st.last_cull_mismatches == 0
            assert_frame_equal(outs[0], ref)
        device.set_option(N.RT_OPT_DIRMASK, 0)
        outs, st = run_gpu(device, scene, size, sky, rnd, jitter, max_objects=slots, variant=variant)
        assert st.dirmask_cells == 0
        assert_frame_equal(outs[0], ref)
    finally:
        device.set_option(N.RT_OPT_DIRMASK, 1)
        device.set_option(N.RT_OPT_DIRMASK_K, 12)
        device.set_option(N.RT_OPT_DIRMASK_CELLS, 4096)
        device.set_option(N.RT_OPT_CULL, 1)


def test_direction_mask_table_is_kept_while_the_spheres_stay(device):
    """A moving camera re-sends the same objects every frame (connection.rs:373-385): the table is rebuilt only when a
    sphere record changes, and frames stay the oracle's through both."""
    scene = cases.headline_scene(4)
    size = (160, 90)
    sky, rnd = cases.inputs(64, 32)
    for step in range(3):
        scene.camera.position = (55.0 + 3.0 * step, 65.0, 55.0)
        if step == 2:
            c = scene.objects[5].geometry.center
            scene.objects[5].geometry.center = (c[0] + 1.5, c[1], c[2] - 0.5)
        outs, st = run_gpu(device, scene, size, sky, rnd, (0.1, 0.1), max_objects="live")
        assert st.dirmask_cells > 0
        assert_frame_equal(outs[0], run_oracle(scene, size, sky, rnd, (0.1, 0.1), max_objects="live"))


@pytest.mark.parametrize("variant", [0, 3])
@pytest.mark.parametrize("case", ["stress_1024", "stress_300", "stress_129", "inside_the_scene", "heap"])
def test_candidate_lists_never_change_a_frame(device, variant, case):
    """rt_dirmask.h "Candidate lists": secondary rays of scenes with more than 128 spheres test only the spheres their table
    entries list, in rounds (a ray whose best hit lies beyond its entry's reach looks the next entry up further along).  The
    frame must be the oracle's (= the linear scan's) bit for bit at several table resolutions and reaches, RT_OPT_CULL = 2
    re-scans every ray linearly inside the kernel and counts disagreements (zero), and the sphere grid (lists off) gives the
    same frame."""
    from rsraytracingv3_amd.scene import Camera, Geometry, Material, Object
    size, jitter = (200, 112), (0.3, -0.2)
    if case == "stress_1024":
        scene = cases.stress_scene(1024, 8)
    elif case == "stress_300":
        scene = cases.stress_scene(300, 6)
        scene.camera = Camera((60.0, 40.0, 60.0), (0.5, -0.8, 0.0), 70.0)
    elif case == "stress_129":
        scene = cases.stress_scene(129, 6)
        scene.camera = Camera((60.0, 40.0, 60.0), (0.5, -0.8, 0.0), 70.0)
    elif case == "inside_the_scene":  # camera between the spheres, looking along the ground: long shallow rays through many entries
        scene = cases.stress_scene(1024, 8)
        scene.camera = Camera((70.0, 2.5, 10.0), (0.02, -1.6, 0.0), 80.0)
    else:  # forty overlapping spheres on one spot: entries without a reach, their rays scan linearly
        scene = cases.stress_scene(200, 6)
        heap = [Object("Heap %d" % i, Material((0.9, 0.5, 0.2), (0.0, 0.0, 0.0), 0.0, 0.5, 0.3 if i % 2 else 0.0),
                       Geometry.sphere((10.0 + 0.01 * i, 4.0, -5.0), 4.0)) for i in range(40)]
        scene.objects = heap + scene.objects
        scene.camera = Camera((40.0, 30.0, 30.0), (0.6, -0.9, 0.0), 70.0)
    sky, rnd = cases.inputs(64, 32)
    ref = run_oracle(scene, size, sky, rnd, jitter, max_objects="live")
    try:
        for k, cells, reach in ((8, 16384, 16), (3, 64, 2), (12, 2048, 64), (6, 100000, 5)):
            device.set_option(N.RT_OPT_DIRLIST, 1)
            device.set_option(N.RT_OPT_DIRLIST_K, k)
            device.set_option(N.RT_OPT_DIRLIST_CELLS, cells)
            device.set_option(N.RT_OPT_DIRLIST_REACH, reach)
            device.set_option(N.RT_OPT_CULL, 2)
            outs, st = run_gpu(device, scene, size, sky, rnd, jitter, max_objects="live", variant=variant)
            assert st.dirmask_cells > 0 and st.dirmask_bins == 6 * k * k and st.dirmask_lists > 0 and st.grid_cells == 0, "the scene got no lists"
            assert st.last_variant == variant and (variant == 3 or st.last_stage == 4)
            assert st.last_cull_mismatches == 0
            assert_frame_equal(outs[0], ref)
        device.set_option(N.RT_OPT_DIRLIST, 0)
        outs, st = run_gpu(device, scene, size, sky, rnd, jitter, max_objects="live", variant=variant)
        assert st.dirmask_lists == 0 and st.grid_cells > 0
        assert_frame_equal(outs[0], ref)
    finally:
        device.set_option(N.RT_OPT_DIRLIST, 1)
        device.set_option(N.RT_OPT_DIRLIST_K, 8)
        device.set_option(N.RT_OPT_DIRLIST_CELLS, 16384)
        device.set_option(N.RT_OPT_DIRLIST_REACH, 16)
        device.set_option(N.RT_OPT_CULL, 1)


@pytest.mark.parametrize("spheres", [1024, 260])
def test_candidate_lists_in_multi_frame_launches_with_a_tail_spill(device, spheres):
    """The headline launch shape on a large scene: render_blocks<.., LISTS> re-packs, hands its last paths to tail_stream<.., LISTS>,
    nothing is staged into LDS.  Texels, floats and totals of a six-frame launch equal the sequential frames' at several spill
    thresholds, and the sequential frames are the oracle's chain."""
    scene = cases.stress_scene(spheres, 8)
    size = (333, 187)
    sky, rnd = cases.inputs(64, 32)
    jit = [(0.1, -0.2), (-0.33, 0.27), (0.45, 0.05), (-0.49, -0.49), (0.0, 0.31), (0.2, 0.2)]
    try:
        device.set_option(N.RT_OPT_FLAGS, 4)  # launches of every size spill
        device.set_option(N.RT_OPT_TAIL_SPILL, 0)
        (seq, bat0), st0 = _batch_vs_sequential(device, scene, size, sky, rnd, jit, 0, N.RT_ACCUM_SRGB8, 3, (0, 1, 8), "live")
        assert st0.last_variant == 3 and st0.last_stage < 100 and st0.dirmask_lists > 0
        assert np.array_equal(seq[0], bat0[0]) and np.array_equal(seq[1].view(np.uint32), bat0[1].view(np.uint32))
        hist = None
        for k in range(len(jit)):
            ref = run_oracle(scene, size, sky, rnd, jit[k], max_objects="live", pc=k, history=hist)
            hist = ref.rgba8
        assert np.array_equal(seq[0], hist), "the sequential frames are not the oracle's chain"
        for t, refill_min, chunk in ((4, 1, 1), (32, 16, 2), (128, 64, 1)):
            device.set_option(N.RT_OPT_TAIL_SPILL, t)
            device.set_option(N.RT_OPT_TAIL_SPILL_SINGLE, t)
            device.set_option(N.RT_OPT_REFILL_MIN, refill_min)
            device.set_option(N.RT_OPT_TILE_CHUNK, chunk)
            (seq_t, bat), st = _batch_vs_sequential(device, scene, size, sky, rnd, jit, 0, N.RT_ACCUM_SRGB8, 3, (0, 1, 8), "live")
            assert st.last_variant == 3 and st.last_stage >= 100, "the launch did not take the spilling shape"
            assert (st.last_segments, st.last_rough_hits) == (st0.last_segments, st0.last_rough_hits)
            assert np.array_equal(seq[0], bat[0]), "threshold %d: texels differ" % t
            assert np.array_equal(seq[1].view(np.uint32), bat[1].view(np.uint32)), "threshold %d: floats differ" % t
            assert np.array_equal(seq[0], seq_t[0])
    finally:
        device.set_option(N.RT_OPT_FLAGS, 0)
        device.set_option(N.RT_OPT_TAIL_SPILL, N.RT_TAIL_SPILL_DEFAULT)
        device.set_option(N.RT_OPT_TAIL_SPILL_SINGLE, N.RT_TAIL_SPILL_SINGLE_DEFAULT)
        device.set_option(N.RT_OPT_REFILL_MIN, 1)
        device.set_option(N.RT_OPT_TILE_CHUNK, 2)
