#!/usr/bin/env python3
"""Multi-rank rehearsal of the tile-parallel path, launched by torch.distributed.run:

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
      --master-port 29533 tools/rehearse_tiles.py [--backend gloo|nccl]

Every rank renders its row strips through the C-ABI (all ranks may share one GPU with
--backend gloo), rank 0 gathers, de-interleaves and compares the frame bit-for-bit with the CPU
oracle, for three pipelined frames (double-buffered send tensors) and then for two launches of five
progressive frames each (rt_render_frames; one gather per launch, history = the previous launch)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="gloo")
    ap.add_argument("--width", type=int, default=331)
    ap.add_argument("--height", type=int, default=187)
    args = ap.parse_args()
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    if args.backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(args.backend)
    import cases
    import oracle
    from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, synth
    from rsraytracingv3_amd import _native as N
    from rsraytracingv3_amd.distributed import TorchTileRenderer
    W, H = args.width, args.height
    scene = cases.headline_scene(6)
    sky, rnd = synth.procedural_skybox(256, 128), synth.random_texture(3)
    dev = Device(local)
    dev.set_outputs(N.RT_OUT_RGBA8)
    RenderTarget.new(dev, (W, H))
    rt = RandomTexture(dev)
    rt.write(rnd)
    conn = Connection(scene, dev, sky, rt)
    conn.update_buffers((W, H), scene, (0.0, 0.0))
    tiles = TorchTileRenderer(dev, W, H, strip_rows=8)
    jit = synth.jitter_sequence(9, 3)
    frames = []
    for k in range(3):  # progressive_count 0 each: independent frames, gather(k-1) beside render(k)
        out = tiles.render(conn, jit[k], 0)
        if out is not None:
            frames.append((k - 1, out.clone()))
    last = tiles.finish()
    if last is not None:
        frames.append((2, last.clone()))
    jit2 = synth.jitter_sequence(11, 10)
    batches = []
    for b in range(2):
        out = tiles.render(conn, jit2[5 * b:5 * b + 5], 5 * b)
        if out is not None:
            batches.append(out.clone())
    last = tiles.finish()
    if last is not None:
        batches.append(last.clone())
    ok = True
    if rank == 0:
        objects = scene.as_bytes(W, H)[0]
        history = None
        for k in range(10):
            ref = oracle.render(objects, 80, scene.config_bytes(W, H), jit2[k], k, sky, rnd, history_rgba8=history,
                                threads=8, want_hits=False)
            history = ref.rgba8
            if k % 5 == 4:
                got = batches[k // 5].cpu().numpy().view(np.uint8).reshape(H, W, 4)
                same = np.array_equal(got, ref.rgba8)
                print("progressive frames %d..%d in one launch: %s" % (k - 4, k, "bit-identical to the oracle" if same else "MISMATCH"))
                ok &= same
        for k, frame in frames:
            ref = oracle.render(objects, 80, scene.config_bytes(W, H), jit[k], 0, sky, rnd, threads=8, want_hits=False)
            got = frame.cpu().numpy().view(np.uint8).reshape(H, W, 4)
            same = np.array_equal(got, ref.rgba8)
            print("frame %d: %s" % (k, "bit-identical to the oracle" if same else "MISMATCH"))
            ok &= same
    flag = torch.tensor([1 if ok else 0])
    dist.broadcast(flag, 0) if args.backend != "nccl" else None
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("REHEARSAL", "OK" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
