#!/usr/bin/env python3
"""RCCL smoke test of the multi-GPU composition on ONE GPU: a world-size-1 nccl process group, the gather
forced through torch.distributed (FrameGatherer.force_collective), TorchTileRenderer's pipeline compared
with plain single-context frames bit for bit.  It exercises every API call the N>1 bench makes (async
gather into unbound views of the stack, work.wait(), index_select) with the real backend."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import cases  # noqa: E402
from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, synth  # noqa: E402
from rsraytracingv3_amd import _native as N  # noqa: E402
from rsraytracingv3_amd.distributed import TorchTileRenderer  # noqa: E402


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    W, H = 640, 360
    scene = cases.headline_scene()
    sky = synth.procedural_skybox(256, 128)
    jit = synth.jitter_sequence(7, 6)
    dev = Device(0)
    dev.set_outputs(N.RT_OUT_RGBA8)
    target = RenderTarget.new(dev, (W, H))
    rt = RandomTexture(dev)
    rt.write(synth.random_texture(3))
    conn = Connection(scene, dev, sky, rt, max_objects="live")
    conn.update_buffers((W, H), scene, (0.0, 0.0))
    want = []
    for k in range(6):
        conn.frame_data.jitter = (float(jit[k, 0]), float(jit[k, 1]))
        conn.frame_data.progressive_count = 0
        conn.draw()
        want.append(target.read_rgba8().copy())
    tiles = TorchTileRenderer(dev, W, H, 8, force_collective=True)
    got = []
    for k in range(6):
        f = tiles.render(conn, jit[k], 0)
        if f is not None:
            got.append(f.cpu().numpy())
    got.append(tiles.finish().cpu().numpy())
    bad = 0
    for k in range(6):
        g = got[k].view(np.uint8).reshape(H, W, 4)
        if not np.array_equal(g, want[k]):
            bad += 1
            print("frame %d differs" % k)
    # two launches of 8 progressive frames each through the same pipeline (one gather per launch)
    jit2 = synth.jitter_sequence(9, 16)
    launches = []
    for b in range(2):
        f = tiles.render(conn, jit2[8 * b:8 * b + 8], 8 * b)
        if f is not None:
            launches.append(f.cpu().numpy())
    launches.append(tiles.finish().cpu().numpy())
    dev.call("rt_bind_outputs", None, None, None)  # back to the context's own target for the reference run
    for k in range(16):
        conn.frame_data.jitter = (float(jit2[k, 0]), float(jit2[k, 1]))
        conn.frame_data.progressive_count = k
        conn.draw()
        if k % 8 == 7:
            if not np.array_equal(launches[k // 8].view(np.uint8).reshape(H, W, 4), target.read_rgba8()):
                bad += 1
                print("8-frame launch %d differs" % (k // 8))
    dev.close()
    dist.destroy_process_group()
    print("nccl single-rank pipeline: %d frames + 2 launches of 8, %d differing" % (6, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if main() else 0)
