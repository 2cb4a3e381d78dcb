#!/usr/bin/env python3
"""rt355 benchmark — BASELINE.json metric: Mrays/s at 1920x1080, 64 spheres, 8 bounces.

A step is one frame (1 spp, progressive_count 0) of BASELINE config 2 through the C-ABI:
64 spheres + the ground plane (65 live objects), 2048x1024 RGBA32F skybox, 600x600 random
texture, all resident in HBM before the timed region.  A ray = one executed iteration of
the bounce loop (frag.hlsl:36-43).  N GPUs split the same frame into row strips
(strong scaling) and gather the sRGB8 texels on rank 0 over RCCL.

Prints ONE JSON line on rank 0.  `roofline` prices the render kernel's ALGORITHMIC bytes
(8 B/pixel + 16 B/segment + 8 B/rough hit) against the 8 TB/s HBM peak, from
per-launch HIP-event durations taken inside the timed region; `valu` prices the algorithmic
flops against the 157.3 TFLOP/s FP32 vector peak (the bound that actually binds — DESIGN.md).
`cpu_baseline` is the CPU oracle (OpenMP) on the same frame, reported, not targeted.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured achievable)
VALU_PEAK_TFLOPS = 157.3    # FP32 vector peak


def algorithmic_bytes(pixels, segments, rough):
    return 8.0 * pixels + 16.0 * segments + 8.0 * rough


def algorithmic_flops(pixels, segments, n_objects):
    return segments * (25.0 * n_objects + 200.0) + 60.0 * pixels


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spheres", type=int, default=64)
    ap.add_argument("--bounces", type=int, default=8)
    ap.add_argument("--slots", type=int, default=0, help="object slots passed (0 = live count; 80 = reference padding)")
    ap.add_argument("--sky", type=int, default=2048, help="skybox width (height = width/2)")
    ap.add_argument("--variant", type=int, default=-1)
    ap.add_argument("--refill-min", type=int, default=1)
    ap.add_argument("--blocks-per-cu", type=int, default=0)
    ap.add_argument("--tile-chunk", type=int, default=2)
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--batch", type=int, default=0,
                    help="consecutive 1-spp frames per kernel launch (rt_render_frames, 1..64); 0 = 8 per GPU of the job, "
                         "so that a launch carries eight frames' worth of tiles whatever the partition")
    ap.add_argument("--single-frame-reference", action="store_true",
                    help="also time 100 frames submitted one per launch (reported in config.one_frame_per_launch)")
    ap.add_argument("--strip-rows", type=int, default=8)
    ap.add_argument("--backend", default="nccl", help="process-group backend (gloo = 1-GPU rehearsal via host)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the frame the CPU baseline renders (0 = all)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import cases
    from rsraytracingv3_amd import Connection, Device, RandomTexture, RenderTarget, synth
    from rsraytracingv3_amd import _native as N

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch N>1 with torch.distributed.run)" % (args.gpus, world))
    if args.steps > 512:
        raise SystemExit("bench.py: --steps must be <= 512 (per-frame event ring)")
    if not 0 <= args.batch <= 64:
        raise SystemExit("bench.py: --batch must be 0 (auto) or 1..64")
    local_rank = local_rank % max(torch.cuda.device_count(), 1)  # rehearsal: several ranks on one GPU
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)

    W, H = args.width, args.height
    scene = cases.spheres_scene(args.spheres, args.bounces, seed=42, request=max(100, int(args.spheres * 1.6))) \
        if args.spheres <= 64 else cases.stress_scene(args.spheres, args.bounces)
    n_objects = len(scene.objects) if args.slots == 0 else args.slots
    sky = synth.procedural_skybox(args.sky, args.sky // 2)
    rnd = synth.random_texture(1)
    jit = synth.jitter_sequence(42, args.steps + args.warmup + 64)

    dev = Device(local_rank)
    dev.set_outputs(N.RT_OUT_RGBA8)
    dev.set_variant(args.variant)
    dev.set_option(N.RT_OPT_REFILL_MIN, args.refill_min)
    dev.set_option(N.RT_OPT_BLOCKS_PER_CU, args.blocks_per_cu)
    dev.set_option(N.RT_OPT_TILE_CHUNK, args.tile_chunk)
    dev.set_option(4, args.flags)
    target = RenderTarget.new(dev, (W, H))
    rt = RandomTexture(dev)
    rt.write(rnd)
    conn = Connection(scene, dev, sky, rt, max_objects=("live" if args.slots == 0 else args.slots))
    conn.update_buffers((W, H), scene, (0.0, 0.0))

    tiles = None
    if world > 1:
        from rsraytracingv3_amd.distributed import TileParallelRenderer
        tiles = TileParallelRenderer(dev, W, H, args.strip_rows)

    # A step is one 1-spp frame of the progressive accumulation (frame k has progressive_count k, as in the
    # reference's loop while the camera rests); --batch B > 1 submits B consecutive frames per launch.
    B = args.batch if args.batch > 0 else min(64, 8 * world)

    def run_frames(first, count):
        """frames first .. first+count-1; returns the number of launches"""
        k, launches = first, 0
        while k < first + count:
            n = min(B, first + count - k)
            if tiles is not None:  # rank 0 gets the previous launch's image assembled while this one renders
                tiles.render(conn, jit[k] if B == 1 else jit[k:k + n], k)
            elif B == 1:
                conn.frame_data.jitter = (float(jit[k, 0]), float(jit[k, 1]))
                conn.frame_data.progressive_count = k
                conn.draw()
            else:
                conn.draw_frames(jit[k:k + n], k)
            k += n
            launches += 1
        return launches

    def fence():
        if tiles is not None:
            tiles.finish()
        dev.sync()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # warm-up in whole launches (W rounded up to a multiple of the batch), so that every launch a profiler
    # sees carries the same number of frames as the timed ones
    warm_frames = -(-args.warmup // B) * B
    run_frames(0, warm_frames)
    fence()
    t0 = time.perf_counter()
    launches = run_frames(warm_frames, args.steps)
    fence()
    elapsed = time.perf_counter() - t0

    # per-launch totals and HIP-event kernel durations of the timed region
    seg = np.zeros(launches, np.uint64)
    rough = np.zeros(launches, np.uint64)
    kms = np.zeros(launches, np.float32)
    dev.call("rt_get_frame_history", launches, seg.ctypes.data_as(C.c_void_p), rough.ctypes.data_as(C.c_void_p),
             kms.ctypes.data_as(C.c_void_p))
    st = dev.stats()
    frame_px = float(st.last_pixels) / max(int(st.last_batch), 1)  # this rank's pixels of one frame
    local = torch.tensor([elapsed, float(seg.sum()), float(rough.sum()), float(kms.sum()), frame_px],
                         dtype=torch.float64, device="cuda")
    if world > 1:
        mx = local.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = local.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed = float(mx[0])
        total_seg, total_rough, total_px = float(sm[1]), float(sm[2]), float(sm[4])
    else:
        total_seg, total_rough, total_px = float(local[1]), float(local[2]), float(local[4])
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    steps = args.steps
    value = total_seg / elapsed / 1e6
    # for reference: the same frames submitted one per launch (the interactive, lowest-latency way)
    single = None
    if world == 1 and B > 1 and args.single_frame_reference:
        n1 = min(100, args.steps)
        B, keep = 1, B
        run_frames(0, 10)
        fence()
        t1 = time.perf_counter()
        run_frames(warm_frames, n1)
        fence()
        dt1 = time.perf_counter() - t1
        B = keep
        seg1 = np.zeros(n1, np.uint64)
        dev.call("rt_get_frame_history", n1, seg1.ctypes.data_as(C.c_void_p), None, None)
        single = {"frames_per_launch": 1, "frames": n1, "ms_per_frame": round(dt1 / n1 * 1e3, 5),
                  "mrays_per_s": round(float(seg1.sum()) / dt1 / 1e6, 2)}
    # roofline of the dominant (only) kernel, rank 0's launches
    k_sec = float(kms.sum()) * 1e-3
    b_alg = algorithmic_bytes(frame_px * steps, float(seg.sum()), float(rough.sum()))
    f_alg = algorithmic_flops(frame_px * steps, float(seg.sum()), n_objects)
    achieved = b_alg / k_sec / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_r01.json")
    if os.path.exists(tpath) and world == 1:
        try:
            with open(tpath) as f:
                traffic = json.load(f).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    out = {
        "metric": "Mrays/s at 1920x1080, 64 spheres, 8 bounces",
        "value": round(value, 2),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "%dx%d, %d spheres + ground plane (%d object slots), %d bounces, 1 spp, %dx%d RGBA32F skybox"
                        % (W, H, args.spheres, n_objects, args.bounces, args.sky, args.sky // 2),
            "partition": "row strips of %d, round-robin over %d GPU(s)%s" % (args.strip_rows, world,
                                                                          ", RCCL gather to rank 0" if world > 1 else ""),
            "kernel_variant": args.variant,
            "frames_per_launch": B,
            "progressive_count": "frame index k (1-spp frames of a resting camera accumulate, app.rs:208-254)",
            "one_frame_per_launch": single,
            "segments_per_pixel": round(total_seg / (float(W) * H * steps), 4),
            "mpixels_per_s": round(W * H * steps / elapsed / 1e6, 2),
        },
        "roofline": {
            "bound": "hbm",
            "achieved": round(achieved, 2),
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 5),
            "traffic": traffic,
            "kernel": {2: "render_persistent", 3: "render_blocks", -1: "auto (render_blocks at this size)"}.get(args.variant, "render_tiles"),
            "avg_launch_ms": round(float(kms.mean()), 5),
            "launches": launches,
            "alg_bytes_per_launch": round(b_alg / launches, 1),
            "note": "VALU-bound path (SURVEY §8d): see 'valu' for the binding roofline",
        },
        "valu": {
            "achieved": round(f_alg / k_sec / 1e12, 3),
            "peak": VALU_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": round(f_alg / k_sec / 1e12 / VALU_PEAK_TFLOPS, 5),
            "alg_flops_per_launch": round(f_alg / launches, 1),
            "kernel_mrays_per_s": round(float(seg.sum()) / k_sec / 1e6, 2),
            "scan_lane_occupancy": round(float(st.last_segments) / (64.0 * st.last_wave_iterations), 4)
            if st.last_wave_iterations else None,
        },
    }
    if world == 1 and not args.no_cpu_baseline:
        import oracle
        rows = (0, H) if args.cpu_rows <= 0 else ((H - args.cpu_rows) // 2, (H - args.cpu_rows) // 2 + args.cpu_rows)
        if args.slots == 0:
            objects, count = scene.live_object_bytes(), len(scene.objects)
        else:
            objects, count = scene.as_bytes(W, H, args.slots)[0], args.slots
        # the GPU box gives one GPU a share of 16 host cores; never use more than that
        threads = max(1, min(oracle.max_threads(), os.cpu_count() or 1, 16))
        cfg = scene.config_bytes(W, H)
        oracle.render(objects, count, cfg, jit[args.warmup], 0, sky, rnd, rows=(rows[0], rows[0] + 8),
                      threads=threads, want_hits=False)  # warm the thread pool
        cseg, cdt, frames, first_seg = 0, 0.0, 0, 0
        first_n = min(B, args.steps)  # frames of the first timed launch
        while (cdt < 12.0 or frames < first_n) and frames < args.steps:  # a bounded sample: whole frames, ~12 s of wall time
            k = warm_frames + frames
            c0 = time.perf_counter()
            fr = oracle.render(objects, count, cfg, jit[k], 0, sky, rnd, rows=rows, threads=threads, want_hits=False)
            cdt += time.perf_counter() - c0
            fs = int(fr.segments[rows[0]:rows[1]].sum())
            first_seg += fs if frames < first_n else 0
            cseg += fs
            frames += 1
        out["cpu_baseline"] = {
            "value": round(cseg / cdt / 1e6, 3),
            "unit": "Mrays/s",
            "cores": threads,
            "kind": "port",
            "sample": "%d frame(s), rows %d..%d of the same %dx%d workload (%d segments, %.1f s wall, %d OpenMP "
                      "threads), CPU oracle oracle/rt_oracle.c; the reference itself cannot run here (no "
                      "Rust/wgpu/Vulkan)" % (frames, rows[0], rows[1], W, H, cseg, cdt, threads),
        }
        # cross-check: the GPU counted the same number of segments for that frame
        out["cpu_baseline"]["segments_match_gpu"] = bool(args.cpu_rows <= 0 and first_seg == int(seg[0]))
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
