"""The multi-GPU frame driven by ONE process (rt_multi_*, include/rt355.h) — the reference's shape: one process, one loop
(app.rs:208-254).  The GPU box has one device, so the rt_multi binds it several times: every context renders its strips,
the peer copies and the assembly on the first context run exactly as they would across devices.  Frames must be the
oracle's, bit for bit, through single frames (several in flight), multi-frame launches and both accumulation modes."""
import ctypes as C

import numpy as np
import pytest

import cases
import oracle
from rsraytracingv3_amd import Connection, Device, MultiDevice, RandomTexture, RenderTarget, synth
from rsraytracingv3_amd import _native as N

pytestmark = pytest.mark.gpu


def oracle_chain(scene, size, sky, rnd, jitters, first_count, accum):
    w, h = size
    objects, count, cfg = scene.live_object_bytes(), len(scene.objects), scene.config_bytes(w, h)
    hist8 = histf = None
    frames = []
    for k, jit in enumerate(jitters):
        ref = oracle.render(objects, count, cfg, (float(jit[0]), float(jit[1])), first_count + k, sky, rnd, history_rgba8=hist8,
                            history_f32=histf, accum_mode=accum, want_hits=False, threads=8)
        hist8, histf = (None, ref.rgbf32) if accum == 1 else (ref.rgba8, None)
        frames.append(ref)
    return frames


@pytest.mark.parametrize("n_dev,strip", [(1, 8), (2, 8), (3, 8), (8, 8), (5, 3)])
def test_one_process_drives_every_device_and_the_frame_is_the_oracles(n_dev, strip):
    scene = cases.headline_scene(4)
    size = (333, 171)  # ragged: the last strip is short, devices own different row counts
    sky, rnd = cases.inputs(128, 64)
    jit = synth.jitter_sequence(9, 16)
    dev = MultiDevice([0] * n_dev, strip)
    try:
        for accum in (N.RT_ACCUM_SRGB8, N.RT_ACCUM_FLOAT32):
            dev.set_accum_mode(accum)
            target = RenderTarget.new(dev, size)
            rt = RandomTexture(dev)
            rt.write(rnd)
            conn = Connection(scene, dev, sky, rt, max_objects="live")
            conn.update_buffers(size, scene, (0.0, 0.0))
            want = oracle_chain(scene, size, sky, rnd, jit[:10], 0, 1 if accum == N.RT_ACCUM_FLOAT32 else 0)
            k = 0
            for launch in (1, 1, 1, 4, 1, 2):  # frames per launch; the single frames overlap on the devices' internal streams
                if launch == 1:
                    conn.frame_data.jitter = (float(jit[k, 0]), float(jit[k, 1]))
                    conn.frame_data.progressive_count = k
                    conn.draw()
                else:
                    conn.draw_frames(jit[k:k + launch], k)
                k += launch
                if launch != 2:  # leave one gather un-read: the next launch overlaps it
                    frame = dev.read_frame_rgba8(*size)
                    assert np.array_equal(frame, want[k - 1].rgba8), "after %d frames, accumulation mode %d" % (k, accum)
            dev.sync()
            assert np.array_equal(target.read_rgba8(), want[k - 1].rgba8)
            st = dev.stats()
            assert st.local_rows == size[1] and st.last_pixels == 2 * size[0] * size[1]  # the last launch carried two frames
            assert st.last_segments == int(want[k - 2].segments.sum()) + int(want[k - 1].segments.sum())
        assert dev._lib.rt_multi_device_count(dev.handle) == n_dev
    finally:
        dev.close()


def test_members_keep_their_partition_and_report_their_own_share():
    scene = cases.headline_scene(3)
    size = (160, 90)
    sky, rnd = cases.inputs(64, 32)
    dev = MultiDevice([0, 0, 0], 8)
    try:
        RenderTarget.new(dev, size)
        rt = RandomTexture(dev)
        rt.write(rnd)
        conn = Connection(scene, dev, sky, rt, max_objects="live")
        conn.update_buffers(size, scene, (0.0, 0.0))
        conn.draw()
        dev.sync()
        lib = N.load()
        rows = []
        for i in range(3):
            member = dev.member(i)
            st = N.RtStats()
            assert lib.rt_get_stats(member, C.byref(st)) == 0
            rows.append(st.local_rows)
            assert lib.rt_set_partition(member, 0, 1, 8) == -1            # the partition belongs to the rt_multi
            assert b"rt_multi" in lib.rt_last_error(member)
        assert rows == [32, 32, 26] and sum(rows) == size[1]                # strips of 8 rows dealt 0, 1, 2, 0, 1, 2, ...
        # a second size: every member's buffers and the first one's frame are replaced
        size2 = (96, 40)
        RenderTarget.new(dev, size2)
        conn.last_size = (0, 0)
        conn.update_buffers(size2, scene, (0.0, 0.0))
        conn.frame_data.progressive_count = 0
        conn.draw()
        ref = oracle.render(scene.live_object_bytes(), len(scene.objects), scene.config_bytes(*size2), conn.frame_data.jitter, 0,
                            sky, rnd, want_hits=False, threads=4)
        assert np.array_equal(dev.read_frame_rgba8(*size2), ref.rgba8)
    finally:
        dev.close()


def test_create_refuses_nonsense():
    lib = N.load()
    handle = C.c_void_p()
    ids = (C.c_int * 1)(0)
    assert lib.rt_multi_create(ids, 0, 8, C.byref(handle)) == -1 and not handle.value
    assert lib.rt_multi_create(ids, 1, 0, C.byref(handle)) == -1
    bad = (C.c_int * 2)(0, 99)
    assert lib.rt_multi_create(bad, 2, 8, C.byref(handle)) == -2 and not handle.value   # RT_ERR_NO_DEVICE, nothing leaked


def test_random_texture_thread_never_splits_a_frame_between_two_textures():
    """rt_multi_set_random from another thread while frames render on three members (random_texture.rs:56-61): every
    assembled frame must be the oracle's frame for ONE of the two textures in ALL of its strips."""
    import threading
    import time
    scene = cases.headline_scene(6)
    sky = synth.procedural_skybox(64, 32)
    tex = [synth.random_texture(11), synth.random_texture(12)]
    size = (96, 54)
    refs = [oracle.render(scene.live_object_bytes(), len(scene.objects), scene.config_bytes(*size), (0.2, -0.3), 0, sky, t,
                          want_hits=False, threads=4).rgba8 for t in tex]
    assert not np.array_equal(refs[0], refs[1])
    dev = MultiDevice([0, 0, 0], 8)
    try:
        RenderTarget.new(dev, size)
        rt = RandomTexture(dev)
        rt.write(tex[0])
        conn = Connection(scene, dev, sky, rt, max_objects="live")
        conn.update_buffers(size, scene, (0.2, -0.3))
        stop, errors = threading.Event(), []

        def writer():
            k = 0
            while not stop.is_set():
                k ^= 1
                rc = dev._lib.rt_set_random(dev.handle, 600, 600, tex[k].ctypes.data_as(C.c_void_p))
                if rc != 0:
                    errors.append(rc)

        th = threading.Thread(target=writer)
        th.start()
        try:
            seen, frames, deadline = [0, 0], 0, time.time() + 20.0
            while (min(seen) < 5 or frames < 60) and time.time() < deadline:
                conn.frame_data.jitter, conn.frame_data.progressive_count = (0.2, -0.3), 0
                conn.draw()
                got = dev.read_frame_rgba8(*size)
                which = [np.array_equal(got, r) for r in refs]
                assert any(which), "the assembled frame matches neither texture: strips rendered with different textures"
                seen[which.index(True)] += 1
                frames += 1
                time.sleep(0.0005)
        finally:
            stop.set()
            th.join()
        assert not errors and min(seen) > 0
    finally:
        dev.close()


def test_more_devices_than_strips():
    """A 20-row frame has three strips of 8 rows: of eight members five own nothing — they launch nothing, send nothing, and
    the frame is still the oracle's."""
    scene = cases.headline_scene(3)
    size = (64, 20)
    sky, rnd = cases.inputs(64, 32)
    dev = MultiDevice([0] * 8, 8)
    try:
        RenderTarget.new(dev, size)
        rt = RandomTexture(dev)
        rt.write(rnd)
        conn = Connection(scene, dev, sky, rt, max_objects="live")
        conn.update_buffers(size, scene, (0.0, 0.0))
        jit = synth.jitter_sequence(4, 4)
        want = oracle_chain(scene, size, sky, rnd, jit, 0, 0)
        conn.frame_data.jitter, conn.frame_data.progressive_count = (float(jit[0, 0]), float(jit[0, 1])), 0
        conn.draw()
        assert np.array_equal(dev.read_frame_rgba8(*size), want[0].rgba8)
        conn.draw_frames(jit[1:4], 1)
        assert np.array_equal(dev.read_frame_rgba8(*size), want[3].rgba8)
        lib = N.load()
        rows = []
        for i in range(8):
            st = N.RtStats()
            assert lib.rt_get_stats(dev.member(i), C.byref(st)) == 0
            rows.append(st.local_rows)
        assert rows == [8, 8, 4, 0, 0, 0, 0, 0]
    finally:
        dev.close()


@pytest.mark.parametrize("members", [1, 3])
def test_asynchronous_presentation_equals_the_synchronous_read_back(members):
    """rt_present_async / rt_present_wait (render_target.rs:50-70 every frame without stalling the loop): frame k's texels are
    copied into a pinned staging slot on a copy stream while frame k + 1 renders; 12 accumulating frames arrive texel for
    texel as a blocking read-back after every frame delivers them, the history (and with it every later frame) is untouched,
    through one context and through an rt_multi's assembled frame."""
    scene = cases.headline_scene(5)
    size = (250, 141)
    sky, rnd = cases.inputs(64, 32)
    jit = synth.jitter_sequence(5, 12)

    def run(asynchronous):
        dev = MultiDevice([0] * members, 8) if members > 1 else Device(0)
        try:
            dev.set_outputs(N.RT_OUT_RGBA8)
            target = RenderTarget.new(dev, size)
            rt = RandomTexture(dev)
            rt.write(rnd)
            conn = Connection(scene, dev, sky, rt, max_objects="live")
            conn.update_buffers(size, scene, (0.0, 0.0))
            frames = []
            for k in range(12):
                conn.frame_data.jitter, conn.frame_data.progressive_count = (float(jit[k, 0]), float(jit[k, 1])), k
                conn.draw()
                if asynchronous:
                    target.present_async(k % 2)          # queue frame k's copy ...
                    if k:
                        frames.append(target.present_wait((k - 1) % 2))  # ... and show frame k - 1 meanwhile
                else:
                    frames.append(dev.read_frame_rgba8(*size) if members > 1 else target.read_rgba8())
            if asynchronous:
                frames.append(target.present_wait(11 % 2))
                with pytest.raises(N.RtError):
                    target.present_wait(0)               # nothing queued into the slot since it was waited for
                with pytest.raises(N.RtError):
                    target.present_async(N.RT355_PRESENT_SLOTS)
            return frames
        finally:
            dev.close()

    sync_frames, async_frames = run(False), run(True)
    assert len(async_frames) == 12
    for k in range(12):
        assert async_frames[k].shape == (size[1], size[0], 4)
        assert np.array_equal(sync_frames[k], async_frames[k]), "frame %d differs" % k
    ref = None
    for k in range(12):
        ref = oracle.render(scene.live_object_bytes(), len(scene.objects), scene.config_bytes(*size), (float(jit[k, 0]), float(jit[k, 1])), k,
                            sky, rnd, history_rgba8=ref.rgba8 if ref else None)
    assert np.array_equal(async_frames[11], ref.rgba8)


def test_peer_status_is_reported_and_two_physical_devices_assemble_the_oracles_frame():
    """rt_multi_peer_status says how the gather reaches the first device.  With one device bound several times nothing crosses a
    link (0 / 0).  Where two physical devices are visible the same frame is rendered across them — the only place the
    hipMemcpyPeerAsync / cross-device event path really runs; this environment's boxes have one GPU, so there the test stops after
    the one-device half (the cross-device path is unverified here: DESIGN.md, multi-GPU)."""
    import torch
    lib = N.load()
    dev = MultiDevice([0, 0, 0], 8)
    try:
        direct, staged = C.c_uint32(7), C.c_uint32(7)
        assert lib.rt_multi_peer_status(dev.handle, C.byref(direct), C.byref(staged)) == 0
        assert (direct.value, staged.value) == (0, 0)
    finally:
        dev.close()
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: the cross-device gather cannot run here")
    scene = cases.headline_scene(4)
    size = (333, 171)
    sky, rnd = cases.inputs(128, 64)
    jit = synth.jitter_sequence(9, 4)
    dev = MultiDevice([0, 1], 8)
    try:
        assert lib.rt_multi_peer_status(dev.handle, C.byref(direct), C.byref(staged)) == 0
        assert direct.value + staged.value == 1
        target = RenderTarget.new(dev, size)
        rt = RandomTexture(dev)
        rt.write(rnd)
        conn = Connection(scene, dev, sky, rt, max_objects="live")
        conn.update_buffers(size, scene, (0.0, 0.0))
        for k in range(4):
            conn.frame_data.jitter, conn.frame_data.progressive_count = (float(jit[k, 0]), float(jit[k, 1])), k
            conn.draw()
        frames = oracle_chain(scene, size, sky, rnd, jit, 0, 0)
        assert np.array_equal(dev.read_frame_rgba8(*size), frames[-1].rgba8)
    finally:
        dev.close()
