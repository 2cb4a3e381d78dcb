"""Full BASELINE sizes on the GPU: properties that need no reference frame (determinism, kernel-shape
equivalence, partition invariance, padding invariance, checksum of segment totals) plus one complete
1920x1080 frame against the oracle."""
import os
import zlib

import numpy as np
import pytest

import cases
import oracle
from rsraytracingv3_amd import Connection, RandomTexture, RenderTarget, render_scene, synth
from rsraytracingv3_amd import _native as N
from rsraytracingv3_amd.partition import assemble_rows

pytestmark = pytest.mark.gpu

SKY = None


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def sky():
    global SKY
    if SKY is None:
        SKY = synth.procedural_skybox(2048, 1024)
    return SKY


def render(device, scene, size, jitter, variant=-1, slots="live", partition=(0, 1, 8), frames=1, outputs=N.RT_OUT_RGBA8):
    device.set_partition(*partition)
    device.set_accum_mode(N.RT_ACCUM_SRGB8)
    device.set_outputs(outputs)
    device.set_variant(variant)
    target = RenderTarget.new(device, size)
    rt = RandomTexture(device)
    rt.write(synth.random_texture(1))
    conn = Connection(scene, device, sky(), rt, max_objects=slots)
    for _ in range(frames):
        render_scene(device, conn, target, scene, jitter)
    device.sync()
    return target, device.stats()


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def test_1080p_headline_frame_equals_oracle(device):
    scene = cases.headline_scene(8)
    size = (1920, 1080)
    target, st = render(device, scene, size, (0.31, -0.17), outputs=N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32 | N.RT_OUT_SEGMENTS)
    ref = oracle.render(scene.live_object_bytes(), len(scene.objects), scene.config_bytes(*size), (0.31, -0.17), 0, sky(),
                        synth.random_texture(1), threads=16, want_hits=False)
    assert st.last_segments == int(ref.segments.sum())
    assert np.array_equal(target.read_segments(), ref.segments)
    got = target.read_rgbf32()
    assert np.nanmax(np.abs(got - ref.rgbf32)) <= 1e-4          # BASELINE.json tolerance
    assert np.array_equal(got.view(np.uint32), ref.rgbf32.view(np.uint32))
    assert np.array_equal(target.read_rgba8(), ref.rgba8)


@pytest.mark.parametrize("size", [(1920, 1080), (3840, 2160)])
def test_kernel_shapes_agree_and_are_deterministic(device, size):
    scene = cases.headline_scene(8)
    sums, segs = [], []
    device.set_option(N.RT_OPT_INFLIGHT, 0)  # the library's rule: a whole frame of 8 bounces renders four in flight (short launches: five)
    for variant in (-1, 0, 1, 2, 3, 3):   # the repeated one checks run-to-run determinism
        target, st = render(device, scene, size, (-0.2, 0.4), variant=variant)
        assert st.last_launches_in_flight == 4
        sums.append(crc(target.read_rgba8()))
        segs.append(st.last_segments)
    assert len(set(sums)) == 1 and len(set(segs)) == 1


def test_partition_reassembles_full_frame_at_1080p(device):
    scene = cases.headline_scene(8)
    size = (1920, 1080)
    whole, st = render(device, scene, size, (0.05, 0.45))
    want = whole.read_rgba8()
    total = 0
    pieces = []
    for rank in range(8):
        t, s = render(device, scene, size, (0.05, 0.45), partition=(rank, 8, 8))
        pieces.append(t.read_rgba8())
        total += s.last_segments
    device.set_partition(0, 1, 8)
    assert total == st.last_segments                      # checksum of the per-rank segment totals
    assert np.array_equal(assemble_rows(pieces, 1080, 8), want)


def test_reference_padding_is_invisible_at_1080p(device):
    scene = cases.headline_scene(8)
    a, sa = render(device, scene, (1920, 1080), (0.0, 0.0), slots=None)      # 80 zero-padded slots
    b, sb = render(device, scene, (1920, 1080), (0.0, 0.0), slots="live")    # 65 live objects
    assert sa.sphere_count == 79 and sb.sphere_count == 64
    assert sa.last_segments == sb.last_segments
    assert crc(a.read_rgba8()) == crc(b.read_rgba8())


def test_stress_1024_spheres_at_1080p_shapes_agree(device):
    """BASELINE config 3 at its full size: the candidate lists in the re-packing shape (the shape its launches take), in the plain
    tile shape and switched off (the sphere grid of round 3) give one and the same frame and segment total; an eight-frame launch
    (one-wave workgroups that spill their tails to tail_stream) equals eight single frames; and a 64-row band of the frame equals
    the oracle's, texel for texel."""
    scene = cases.stress_scene(1024, 8)
    size, jitter = (1920, 1080), (0.1, 0.1)
    sums = {}
    try:
        for name, variant, lists in (("lists, re-packing", 3, 1), ("lists, tiles", 0, 1), ("grid", -1, 0)):
            device.set_option(N.RT_OPT_DIRLIST, lists)
            t, st = render(device, scene, size, jitter, variant=variant)
            assert st.sphere_count == 1024 and (st.dirmask_lists > 0) == bool(lists) and (st.grid_cells > 0) != bool(lists)
            sums[name] = (crc(t.read_rgba8()), st.last_segments)
            if name == "lists, re-packing":
                band = t.read_rgba8(rows=(512, 64))
        assert sums["lists, re-packing"] == sums["lists, tiles"] == sums["grid"], sums
    finally:
        device.set_option(N.RT_OPT_DIRLIST, 1)
    ref = oracle.render(scene.live_object_bytes(), len(scene.objects), scene.config_bytes(*size), jitter, 0, sky(), synth.random_texture(1),
                        rows=(512, 576), threads=8, want_hits=False)
    assert np.array_equal(band, ref.rgba8[512:576])
    # eight frames in one launch against eight launches
    device.set_outputs(N.RT_OUT_RGBA8)
    device.set_variant(-1)
    jit = synth.jitter_sequence(3, 8)
    results = []
    for batched in (False, True):
        target = RenderTarget.new(device, size)
        rt = RandomTexture(device)
        rt.write(synth.random_texture(1))
        conn = Connection(scene, device, sky(), rt, max_objects="live")
        conn.update_buffers(size, scene, (0.0, 0.0))
        if batched:
            conn.draw_frames(jit, 0)
        else:
            for k in range(8):
                conn.frame_data.jitter, conn.frame_data.progressive_count = (float(jit[k, 0]), float(jit[k, 1])), k
                conn.draw()
        device.sync()
        st = device.stats()
        results.append(crc(target.read_rgba8()))
        if batched:
            assert st.last_variant == 3 and st.last_stage >= 100 and st.last_batch == 8, "the launch did not spill its tail"
    assert results[0] == results[1]


def test_accumulation_converges_at_1080p(device):
    """Progressive accumulation through the sRGB8 history: later frames move the image less and less
    (opacity 1/(n+1), frag.hlsl:61-62)."""
    scene = cases.headline_scene(4)
    size = (1920, 1080)
    device.set_partition(0, 1, 8)
    device.set_outputs(N.RT_OUT_RGBA8)
    device.set_variant(-1)
    target = RenderTarget.new(device, size)
    rt = RandomTexture(device)
    rt.write(synth.random_texture(1))
    conn = Connection(scene, device, sky(), rt, max_objects="live")
    prev, deltas = None, []
    for k in range(12):
        render_scene(device, conn, target, scene)          # seeded jitter stream, progressive_count = k
        device.sync()
        assert conn.frame_data.progressive_count == k
        img = target.read_rgba8()[..., :3].astype(np.int16)
        if prev is not None:
            deltas.append(float(np.abs(img - prev).mean()))
        prev = img
    assert deltas[-1] < 0.35 * deltas[0]


def test_rccl_pipeline_single_rank():
    """tools/nccl_single.py: the N>1 composition (async RCCL gather, double-buffered strips, de-interleave)
    through a world-size-1 nccl group on this GPU, frames bit-identical to plain rendering.  Own process so
    the process group does not leak into the other tests."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "nccl_single.py")], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "0 differing" in p.stdout


def test_tile_parallel_two_ranks_sharing_this_gpu():
    """tools/rehearse_tiles.py under torch.distributed.run with 2 ranks on this one GPU (gather staged through
    gloo): single frames and 5-frame launches, strips gathered and de-interleaved on rank 0, bit-identical to
    the oracle's frames.  The N > 1 composition short of RCCL itself (which needs one GPU per rank)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "tools", "rehearse_tiles.py"), "--backend", "gloo"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "REHEARSAL OK" in p.stdout


def test_comm_path_of_the_c_abi_world_1():
    """tools/comm_rehearsal.py: rt_comm_get_unique_id / rt_comm_init / the RCCL gather inside librt355 / rt_read_frame_rgba8
    with a one-rank communicator on this GPU — single frames (two in flight), multi-frame launches, both accumulation
    modes, a ragged frame size — bit-identical to the oracle's frame chain.  Own process (RCCL state)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run([sys.executable, os.path.join(root, "tools", "comm_rehearsal.py")], capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "COMM OK: world 1" in p.stdout


@pytest.mark.parametrize("world,strip", [(2, 8), (3, 8), (8, 8), (5, 3)])
def test_rank0_assembly_of_a_tile_split_frame(device, world, strip):
    """What rank 0 does with the strips it receives (rt_assemble_strips = the gather's own layout function and
    kernel, fed from the host): every rank's strips rendered with rt_set_partition on this GPU, the peers' rows
    concatenated in rank order as the grouped ncclRecv delivers them -> the single-GPU frame, bit for bit.  A ragged
    height, so that ranks own different row counts and the last strip is short."""
    import ctypes as C
    scene = cases.headline_scene(4)
    size = (203, 117)
    full = render(device, scene, size, (0.2, -0.3))[0].read_rgba8()
    pieces = [render(device, scene, size, (0.2, -0.3), partition=(r, world, strip))[0].read_rgba8() for r in range(world)]
    device.set_partition(0, 1, 8)
    assert sum(p.shape[0] for p in pieces) == size[1]
    peers = np.ascontiguousarray(np.concatenate(pieces[1:], axis=0)) if world > 1 else np.zeros((1, size[0], 4), np.uint8)
    own = np.ascontiguousarray(pieces[0])
    out = np.zeros((size[1], size[0], 4), np.uint8)
    device.call("rt_assemble_strips", size[0], size[1], world, strip, own.ctypes.data_as(C.c_void_p),
                peers.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p))
    assert np.array_equal(out, full)
    assert np.array_equal(out, assemble_rows(pieces, size[1], strip))  # the host-side mirror agrees


def test_a_launch_of_more_than_2_to_the_26_pixels_keeps_the_re_packing_shape(device):
    """40 frames of 1920x1080 in one launch are 83 M pixels: the re-packing shape used to keep 26 bits of pixel index in its
    exchange record and fell back to plain tiles beyond 67 M.  Same texels as five launches of eight frames."""
    scene = cases.headline_scene(8)
    size = (1920, 1080)
    jit = synth.jitter_sequence(7, 40)
    device.set_partition(0, 1, 8)
    device.set_accum_mode(N.RT_ACCUM_SRGB8)
    device.set_outputs(N.RT_OUT_RGBA8)
    device.set_variant(-1)
    rt = RandomTexture(device)
    rt.write(synth.random_texture(1))
    sums = []
    for per_launch in (40, 8):
        target = RenderTarget.new(device, size)
        conn = Connection(scene, device, sky(), rt, max_objects="live")
        conn.update_buffers(size, scene, (0.0, 0.0))
        for k in range(0, 40, per_launch):
            conn.draw_frames(jit[k:k + per_launch], k)
        device.sync()
        st = device.stats()
        assert st.last_variant == 3 and st.last_batch == per_launch
        sums.append(crc(target.read_rgba8()))
    assert sums[0] == sums[1]
