"""BASELINE config 5 on the GPU: 3840x2160, 64 spheres, 8 bounces, 256 accumulated frames against an
8192x4096 (512 MiB — larger than the 256 MiB Infinity Cache) skybox loaded from an OpenEXR file.

What the reference does on this path: `Connection::load_hdri` decodes the 8k panorama and uploads it
(connection.rs:176-215), every loop turn renders one 1-spp frame whose result is mixed into the stored
Rgba8UnormSrgb frame with weight 1/(progressive_count+1) (frag.hlsl:60-62, connection.rs:385-387), and the
random texture is replaced every 4 ms by another thread (random_texture.rs:56-61).

Checked here, through the C-ABI:
  * the 8k skybox goes PIZ .exr -> rt_set_skybox_exr (device) and PIZ .exr -> rt_exr_read_rgba32f (the host
    array the oracle samples) — the file is written by tests/exr_writer.py in a child process;
  * all 256 frames, both accumulation modes (reference sRGB8 feedback, float32 history), random texture
    regenerated on the device every 8 frames: a 64-row band of every frame — float value and stored texel —
    is bit-identical to the oracle's frame chain on the same band (the oracle takes `rows=`);
  * the whole 3840x2160 frame after 256 single-frame launches (rt_render) has the same CRC as after 32
    eight-frame launches (rt_render_frames), in both modes.
Oracle status: CPU restatement, parity unpinned (oracle/rt_oracle.c header)."""
import os
import subprocess
import sys
import zlib

import numpy as np
import pytest

import cases
import oracle
from rsraytracingv3_amd import Connection, RandomTexture, RenderTarget, synth
from rsraytracingv3_amd import _native as N

pytestmark = pytest.mark.gpu

W, H, FRAMES, REFRESH = 3840, 2160, 256, 8
SKY_W = 8192
BAND = (1016, 64)   # rows 1016..1079, mid-frame: spheres, the ground, and the skybox through every reflected miss
RND_SEED = 500


@pytest.fixture(scope="module")
def skybox_exr(tmp_path_factory):
    """The 8192x4096 procedural panorama as a PIZ-compressed HALF .exr, written by a child process (pure
    Python encoder on every host core; this process holds the GPU and must not fork a pool)."""
    path = str(tmp_path_factory.mktemp("config5") / "sky8k_piz.exr")
    here = os.path.dirname(os.path.abspath(__file__))
    workers = str(max(1, min(os.cpu_count() or 1, 16)))
    p = subprocess.run([sys.executable, os.path.join(here, "exr_writer.py"), "skybox", str(SKY_W), path, "piz", workers],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-1000:] + p.stderr[-2000:]
    return path


def _accumulate(device, conn, target, rt, jit, frames_per_launch, on_frame=None):
    """256 frames of the accumulation; the random texture is regenerated on the device before frames 0, 8, ...
    (stream seeds RND_SEED, RND_SEED+1, ...).  on_frame(k) runs after every single-frame launch."""
    rt.seed = RND_SEED
    k = 0
    while k < FRAMES:
        if k % REFRESH == 0:
            rt.generate_on_device()
        if frames_per_launch == 1:
            conn.frame_data.jitter = (float(jit[k, 0]), float(jit[k, 1]))
            conn.frame_data.progressive_count = k
            conn.draw()
            if on_frame is not None:
                on_frame(k)
        else:
            assert REFRESH % frames_per_launch == 0  # the texture must not change inside a launch
            conn.draw_frames(jit[k:k + frames_per_launch], k)
        k += frames_per_launch
    device.sync()


@pytest.mark.parametrize("accum", [N.RT_ACCUM_SRGB8, N.RT_ACCUM_FLOAT32], ids=["srgb8_feedback", "float32_history"])
def test_config5_4k_256_frames_8k_exr_skybox(device, skybox_exr, accum):
    scene = cases.headline_scene(8)
    jit = synth.jitter_sequence(5, FRAMES)
    (sw, sh), sky = Connection.load_image(skybox_exr)  # host decode: what the oracle samples
    assert (sw, sh) == (SKY_W, SKY_W // 2) and sky.nbytes == 512 << 20
    expect = synth.procedural_skybox(SKY_W, SKY_W // 2)[..., :3].astype(np.float16).astype(np.float32)
    assert np.array_equal(sky[..., :3].view(np.uint32), expect.view(np.uint32)) and np.all(sky[..., 3] == 1.0)
    del expect

    device.set_partition(0, 1, 8)
    device.set_variant(-1)
    device.set_accum_mode(accum)
    device.set_outputs(N.RT_OUT_RGBA8 | N.RT_OUT_RGBF32)
    target = RenderTarget.new(device, (W, H))
    rt = RandomTexture(device, seed=RND_SEED)
    rt.generate_on_device()
    conn = Connection(scene, device, skybox_exr, rt, max_objects="live")  # load_hdri(path) = rt_set_skybox_exr
    conn.update_buffers((W, H), scene, (0.0, 0.0))

    objects, count, cfg = scene.live_object_bytes(), len(scene.objects), scene.config_bytes(W, H)
    threads = max(1, min(oracle.max_threads(), os.cpu_count() or 1, 16))
    y0, y1 = BAND[0], BAND[0] + BAND[1]
    state = {"hist8": None, "histf": None, "rnd": None, "bad": [], "segments": 0}

    def check_frame(k):
        if k % REFRESH == 0:  # the host twin of the device stream (test_edge_cases_gpu pins them bit for bit)
            state["rnd"] = synth.random_texture(RND_SEED + k // REFRESH)
        ref = oracle.render(objects, count, cfg, jit[k], k, sky, state["rnd"], history_rgba8=state["hist8"],
                            history_f32=state["histf"], accum_mode=1 if accum == N.RT_ACCUM_FLOAT32 else 0,
                            rows=(y0, y1), threads=threads, want_hits=False)
        state["segments"] += int(ref.segments[y0:y1].sum())
        if accum == N.RT_ACCUM_FLOAT32:
            state["histf"] = ref.rgbf32
        else:
            state["hist8"] = ref.rgba8
        got_f = target.read_rgbf32(rows=BAND)
        got_8 = target.read_rgba8(rows=BAND)
        if not (np.array_equal(got_f.view(np.uint32), ref.rgbf32[y0:y1].view(np.uint32)) and
                np.array_equal(got_8, ref.rgba8[y0:y1])):
            state["bad"].append(k)

    _accumulate(device, conn, target, rt, jit, 1, on_frame=check_frame)
    assert state["bad"] == [], "frames whose band differs from the oracle's chain: %s" % state["bad"][:16]
    assert state["segments"] > 2 * BAND[1] * W * FRAMES  # the band is not a sky-only strip
    seq_texels = target.read_rgba8()
    seq_float_band = target.read_rgbf32(rows=BAND)
    crc_seq = zlib.crc32(seq_texels.tobytes())
    assert seq_texels[..., 3].min() == 255
    del seq_texels

    # the same accumulation as 32 launches of 8 frames
    target = RenderTarget.new(device, (W, H))  # fresh target + history
    conn.last_size = (0, 0)
    conn.update_buffers((W, H), scene, (0.0, 0.0))
    _accumulate(device, conn, target, rt, jit, REFRESH)
    batch_texels = target.read_rgba8()
    assert zlib.crc32(batch_texels.tobytes()) == crc_seq
    assert np.array_equal(target.read_rgbf32(rows=BAND).view(np.uint32), seq_float_band.view(np.uint32))
    ref8 = state["hist8"] if accum == N.RT_ACCUM_SRGB8 else None
    if ref8 is not None:
        assert np.array_equal(batch_texels[y0:y1], ref8[y0:y1])
