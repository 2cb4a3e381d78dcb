#!/usr/bin/env python3
"""The multi-GPU path of the C-ABI (rt_comm_init: RCCL gather inside librt355) with WORLD_SIZE ranks.

  python tools/comm_rehearsal.py                              world 1 on this GPU (always possible)
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/comm_rehearsal.py
                                                             one rank per GPU (ranks sharing a GPU: RCCL refuses)
The unique id travels through a gloo group (host memory); every rank renders its strips of single frames (two in
flight), of multi-frame launches, in both accumulation modes; rank 0 compares the assembled frames with the
oracle's frame chain, bit for bit.  Prints COMM OK."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import cases  # noqa: E402
import oracle  # noqa: E402
from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, synth  # noqa: E402
from rsraytracingv3_amd import _native as N  # noqa: E402

world = int(os.environ.get("WORLD_SIZE", "1"))
rank = int(os.environ.get("RANK", "0"))
local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
if world > 1:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo")
usable, rccl_text = Device.comm_available()  # not a collective: every rank asks before anyone joins
if not usable:
    print("rank %d: no RCCL (%s)" % (rank, rccl_text), flush=True)
    sys.exit(2)
ident = [Device.comm_unique_id() if rank == 0 else None]
if world > 1:
    dist.broadcast_object_list(ident, src=0)
dev = Device(local)
dev.comm_init(ident[0], rank, world, 8)
W, H = 333, 171  # ragged: the last strip is short, ranks own different row counts
scene = cases.headline_scene(4)
sky, rnd = cases.inputs(128, 64)
jit = synth.jitter_sequence(3, 16)
objects, count, cfg = scene.live_object_bytes(), len(scene.objects), scene.config_bytes(W, H)
ok = True
for accum in (N.RT_ACCUM_SRGB8, N.RT_ACCUM_FLOAT32):
    dev.set_accum_mode(accum)
    dev.set_outputs(N.RT_OUT_RGBA8)
    target = RenderTarget.new(dev, (W, H))
    rt = RandomTexture(dev)
    rt.write(rnd)
    conn = Connection(scene, dev, sky, rt, max_objects="live")
    conn.update_buffers((W, H), scene, (0.0, 0.0))
    hist8 = histf = None
    k = 0
    for launch in (1, 1, 1, 4, 1, 3, 1):  # frames per launch
        if launch == 1:
            conn.frame_data.jitter = (float(jit[k, 0]), float(jit[k, 1]))
            conn.frame_data.progressive_count = k
            conn.draw()
        else:
            conn.draw_frames(jit[k:k + launch], k)
        for f in range(launch):
            ref = oracle.render(objects, count, cfg, jit[k + f], k + f, sky, rnd, history_rgba8=hist8, history_f32=histf,
                                accum_mode=1 if accum == N.RT_ACCUM_FLOAT32 else 0, want_hits=False)
            hist8, histf = (None, ref.rgbf32) if accum == N.RT_ACCUM_FLOAT32 else (ref.rgba8, None)
        k += launch
        if rank == 0 and launch != 3:  # also leave some gathers un-read: the next launch overlaps them
            frame = dev.read_frame_rgba8(W, H)
            if not np.array_equal(frame, ref.rgba8):
                ok = False
                print("rank 0: assembled frame after %d frames differs in %d texels (accum %d)"
                      % (k, int((frame != ref.rgba8).any(-1).sum()), accum), flush=True)
    dev.sync()
    if rank == 0:
        frame = dev.read_frame_rgba8(W, H)
        ok = ok and np.array_equal(frame, ref.rgba8)
# With a communicator the two targets alternate: rt_get_device_buffers must name the one the last launch wrote, and
# rt_write_history_rgba8 the one the next launch reads (waiting for the gather that may still send from it).
import ctypes as C  # noqa: E402
hip = C.CDLL(None)  # the HIP runtime the library runs on is already mapped
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
dev.set_accum_mode(N.RT_ACCUM_SRGB8)
target = RenderTarget.new(dev, (W, H))
conn.last_size = (0, 0)
conn.update_buffers((W, H), scene, (0.0, 0.0))
rows = dev.stats().local_rows
seen = set()
for k in range(3):
    conn.frame_data.jitter, conn.frame_data.progressive_count = (float(jit[k, 0]), float(jit[k, 1])), 0
    conn.draw()  # no sync: the gather of this launch is still in flight when the history is overwritten below
    seen.add(dev.device_buffers().rgba8)
    resume = np.random.default_rng(100 + k + 10 * rank).integers(0, 256, (rows, W, 4), dtype=np.uint8)
    resume[..., 3] = 255
    dev.call("rt_write_history_rgba8", resume.ctypes.data_as(C.c_void_p), resume.nbytes)
    conn.frame_data.jitter, conn.frame_data.progressive_count = (float(jit[k + 1, 0]), float(jit[k + 1, 1])), 3
    conn.draw()
    dev.sync()
    bufs = dev.device_buffers()
    seen.add(bufs.rgba8)
    direct = np.empty((rows, W, 4), np.uint8)
    if hip.hipMemcpy(direct.ctypes.data_as(C.c_void_p), bufs.rgba8, direct.nbytes, 2) != 0:
        ok = False
    mine = target.read_rgba8()
    if not np.array_equal(direct, mine):
        ok = False
        print("rank %d: rt_get_device_buffers does not name the current target (launch %d)" % (rank, k), flush=True)
    if world == 1:  # the oracle's frame on top of that history
        want = oracle.render(objects, count, cfg, jit[k + 1], 3, sky, rnd, history_rgba8=resume, want_hits=False).rgba8
        if not np.array_equal(mine, want):
            ok = False
            print("rt_write_history_rgba8 with a communicator: %d texels differ" % int((mine != want).any(-1).sum()), flush=True)
if len(seen) != 2:
    ok = False
    print("rank %d: the targets did not alternate (%d distinct)" % (rank, len(seen)), flush=True)
# the watchdog calls: everything queued has completed after a sync; after rt_comm_abort the context is a single-GPU
# context again and renders the whole frame by itself
if not dev.comm_idle():
    ok = False
    print("rank %d: rt_comm_idle says busy after rt_sync" % rank, flush=True)
if world > 1:
    flag = [ok]
    dist.broadcast_object_list(flag, src=0)
    ok = flag[0]
    dist.barrier()
if world == 1:
    dev.comm_abort()
    target = RenderTarget.new(dev, (W, H))
    conn.last_size = (0, 0)
    conn.update_buffers((W, H), scene, (0.0, 0.0))
    conn.frame_data.jitter, conn.frame_data.progressive_count = (float(jit[0, 0]), float(jit[0, 1])), 0
    conn.draw()
    dev.sync()
    alone = oracle.render(objects, count, cfg, jit[0], 0, sky, rnd, want_hits=False)
    if not np.array_equal(target.read_rgba8(), alone.rgba8):
        ok = False
        print("after rt_comm_abort the single-GPU frame differs from the oracle", flush=True)
else:
    dev.comm_destroy()
dev.close()
if world > 1:
    dist.destroy_process_group()
if rank == 0:
    print("COMM OK: world %d, frames bit-identical to the oracle's chain" % world if ok else "COMM FAILED", flush=True)
sys.exit(0 if ok else 1)
