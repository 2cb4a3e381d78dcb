"""Committed golden frames (tests/golden/*.npz, made by tests/golden/make_golden.py with the
CPU oracle).  CPU tier: the oracle still reproduces them bit for bit (freezes the arithmetic
specification).  GPU tier: the HIP path reproduces them through the C-ABI."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_golden  # noqa: E402


def load(name):
    return np.load(os.path.join(HERE, "golden", name + ".npz"))


@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_oracle_reproduces_golden(name):
    gold = load(name)
    out, _ = make_golden.render_case(name)
    assert np.array_equal(out["input_crc"], gold["input_crc"]), "synthetic inputs changed"
    for key in gold.files:
        a, b = out[key], gold[key]
        assert a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8)), key


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(make_golden.CASES))
def test_hip_reproduces_golden(device, name):
    from rsraytracingv3_amd import Connection, RandomTexture, RenderTarget, render_scene
    from rsraytracingv3_amd import _native as N
    gold = load(name)
    factory, size, sky_size, jitters, slots = make_golden.CASES[name]
    import cases
    scene = factory()
    sky, rnd = cases.inputs(*sky_size)
    device.set_partition(0, 1, 8)
    device.set_accum_mode(N.RT_ACCUM_SRGB8)
    device.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32 | N.RT_OUT_SEGMENTS)
    device.set_variant(-1)
    target = RenderTarget.new(device, size)
    rt = RandomTexture(device)
    rt.write(rnd)
    conn = Connection(scene, device, sky, rt, max_objects=slots)
    for k, j in enumerate(jitters):
        render_scene(device, conn, target, scene, j)
        device.sync()
        assert conn.frame_data.progressive_count == k
        got = target.read_rgbf32()
        assert np.abs(got - gold["rgbf32_%d" % k]).max() <= 1e-4   # BASELINE.json tolerance
        assert np.array_equal(got.view(np.uint32), gold["rgbf32_%d" % k].view(np.uint32))
        assert np.array_equal(target.read_rgba8(), gold["rgba8_%d" % k])
        assert np.array_equal(target.read_segments(), gold["segments_%d" % k])
