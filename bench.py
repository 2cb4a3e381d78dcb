#!/usr/bin/env python3
"""rt355 benchmark — BASELINE.json metric: Mrays/s at 1920x1080, 64 spheres, 8 bounces.

A step is one pass of the hot path over one batch of synthetic input: ONE LAUNCH of 8 x N consecutive 1-spp frames of
BASELINE config 2 through the C-ABI (rt_render_frames) on N GPUs — 64 spheres + the ground plane (65 live objects),
2048x1024 RGBA32F skybox, 600x600 random texture, everything resident in HBM before the timed region.  The frames are
the progressive accumulation of a resting camera (frame k has progressive_count k, app.rs:208-254 /
connection.rs:386-388).  A ray = one executed iteration of the bounce loop (frag.hlsl:36-43).  N GPUs split every
frame into row strips and gather the sRGB8 texels on rank 0 after every launch, so a step gives every GPU eight frames'
worth of pixels whatever N is ("scaling": "weak"); `config.frames_per_step` says how many frames a step carried.
(Until round 2 a step was one frame and a block of --steps 20 at N = 8 one launch of 20 eighth-frames: a third of a
millisecond between two barriers.  `config.block_of_frames` still reports that protocol.)

Protocol: an untimed clock warm-up of >= 2 s of back-to-back launches (on top of the
--warmup steps), then the block of EXACTLY --steps steps — bracketed by barrier + synchronize on both sides,
MAX over ranks — is repeated until >= 1 s has been measured and the MEDIAN block is reported
(`steps`, `ms_per_step` keep their meaning: ms_per_step = median block / steps).

`value` is the headline leg: 8 frames per launch per GPU.  The same JSON line always carries, under `config`, the
one-frame-per-launch leg (the reference's call pattern: one draw per loop turn; there a step is one frame) and the
reference-exact 80-slot leg (15 zero-padded objects scanned as the reference's shader does).

`roofline` is the bound that binds: `bound` "valu" — the render kernels' ALGORITHMIC flops (25 per
object a segment would test in the reference's scan + 200 per segment + 60 per pixel) against the 157.3 TFLOP/s
FP32 vector peak, from per-launch HIP-event durations taken on the context's stream inside the timed region.  That
fraction is in the reference's units: the candidate tables skip most of the scan it prices, so it can exceed 1
(`frac` is null then, the labelled field keeps the figure); `issue_util` is the hardware's own fraction (vector
instructions x 2 cycles over the SIMDs' busy cycles), imported with `traffic` from the PMC profile of the same
sources under profiles/.  `roofline.hbm` keeps the HBM pricing (8 B/pixel + 16 B/segment + 8 B/rough hit against
8 TB/s).  `cpu_baseline` is the CPU oracle (OpenMP) on the same frames, reported,
not targeted.
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
SIMDS = 1024                # 256 CUs x 4
CLOCK_WARMUP_S = 2.0
MIN_TIMED_S = 1.0
KERNEL_NAMES = {0: "render_tiles", 1: "render_tiles<0>", 2: "render_persistent", 3: "render_blocks",
                4: "primary_stream<4> + bounce_stream<2, 6>"}


def kernel_name(st):
    """The render kernel the last launch used, from rt_stats (launch shape + what it staged into LDS)."""
    v = int(st.last_variant)
    if v == 0:
        return "render_tiles<%d>%s" % (st.last_stage, " (sphere grid)" if st.last_stage == 3 else "")
    if v == 3:
        stage = st.last_stage % 100
        return "render_blocks<%d, %d>%s" % (stage // 10, stage % 10, " + tail_stream" if st.last_stage >= 100 else "")
    return KERNEL_NAMES.get(v, "?")


def algorithmic_bytes(pixels, segments, rough):
    return 8.0 * pixels + 16.0 * segments + 8.0 * rough


def algorithmic_flops(pixels, segments, n_objects):
    return segments * (25.0 * n_objects + 200.0) + 60.0 * pixels


def usable_cores():
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                fields = f.read().split()
            quota = float(fields[0]) if fields[0] != "max" else -1.0
            period = float(fields[1]) if len(fields) > 1 else 100000.0
            if path.endswith("cfs_quota_us"):
                with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                    period = float(f.read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


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
    ap.add_argument("--batch-inflight", type=int, default=0,
                    help="RT_OPT_BATCH_INFLIGHT: multi-frame launches in flight (0 = the library's rule: four while their sample buffers "
                         "fit 2 GiB, the next launches render while this one's tail drains and resolves; 1 = one at a time: kernel "
                         "times do not overlap)")
    ap.add_argument("--inflight", type=int, default=0, help="RT_OPT_INFLIGHT: single-frame launches in flight (0 = library default)")
    ap.add_argument("--grid-min", type=int, default=0, help="RT_OPT_GRID_MIN: spheres from which a scene gets the sphere grid (0 = library default 129)")
    ap.add_argument("--dirmask", type=int, default=-1, help="RT_OPT_DIRMASK: 0 = secondary rays scan every sphere, 1 = direction-mask table (library default)")
    ap.add_argument("--dirmask-k", type=int, default=0, help="RT_OPT_DIRMASK_K (0 = library default)")
    ap.add_argument("--dirmask-cells", type=int, default=0, help="RT_OPT_DIRMASK_CELLS (0 = library default)")
    ap.add_argument("--dirlist", type=int, default=-1, help="RT_OPT_DIRLIST: candidate lists of scenes with more than 128 spheres (0 = the sphere grid, 1 = library default)")
    ap.add_argument("--dirlist-min", type=int, default=0, help="RT_OPT_DIRLIST_MIN: fewest spheres that get candidate lists (0 = library default 129)")
    ap.add_argument("--dirlist-k", type=int, default=0, help="RT_OPT_DIRLIST_K (0 = library default)")
    ap.add_argument("--dirlist-cells", type=int, default=0, help="RT_OPT_DIRLIST_CELLS (0 = library default)")
    ap.add_argument("--dirlist-reach", type=int, default=0, help="RT_OPT_DIRLIST_REACH (0 = library default)")
    ap.add_argument("--tail-spill", type=int, default=-1,
                    help="RT_OPT_TAIL_SPILL: workgroups with at most this many paths left hand them to the tail kernel (0 = off, -1 = library default)")
    ap.add_argument("--tail-spill-single", type=int, default=-1, help="RT_OPT_TAIL_SPILL_SINGLE: the same for single-frame launches (-1 = library default 32)")
    ap.add_argument("--batch", type=int, default=0,
                    help="consecutive 1-spp frames per kernel launch (rt_render_frames, 1..64); 0 = 8 per GPU of the job, "
                         "so that a launch carries eight frames' worth of tiles whatever the partition")
    ap.add_argument("--strip-rows", type=int, default=8)
    ap.add_argument("--backend", default="nccl", help="process-group backend (gloo = 1-GPU rehearsal via host)")
    ap.add_argument("--force-comm", action="store_true",
                    help="rehearsal on one GPU: run the N = 1 job through a one-rank communicator (rt_comm_init), as N > 1 does")
    ap.add_argument("--comm-timeout", type=float, default=90.0,
                    help="N > 1: seconds the pre-flight launch's RCCL gather may take before the bench falls back to torch.distributed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the one-frame-per-launch and 80-slot legs (profiling)")
    ap.add_argument("--min-seconds", type=float, default=MIN_TIMED_S, help="minimum timed window of a leg")
    ap.add_argument("--clock-warmup", type=float, default=CLOCK_WARMUP_S, help="seconds of untimed launches before timing")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the frame the CPU baseline renders (0 = all)")
    args = ap.parse_args()

    # stdout carries ONE JSON line: native libraries that print there (RCCL's version banner at communicator creation)
    # are pointed at stderr for the duration of the run; the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

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
    if args.steps < 1 or args.steps > 512:
        raise SystemExit("bench.py: --steps must be 1..512 (per-launch event ring)")
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
    sky = synth.procedural_skybox(args.sky, args.sky // 2)
    rnd = synth.random_texture(1)
    n_jit = 4096
    jit = synth.jitter_sequence(42, n_jit)

    dev = Device(local_rank)
    dev.set_outputs(N.RT_OUT_RGBA8)
    dev.set_variant(args.variant)
    dev.set_option(N.RT_OPT_REFILL_MIN, args.refill_min)
    dev.set_option(N.RT_OPT_BLOCKS_PER_CU, args.blocks_per_cu)
    dev.set_option(N.RT_OPT_TILE_CHUNK, args.tile_chunk)
    dev.set_option(N.RT_OPT_FLAGS, args.flags)
    dev.set_option(N.RT_OPT_BATCH_INFLIGHT, args.batch_inflight)
    if args.inflight > 0:
        dev.set_option(N.RT_OPT_INFLIGHT, args.inflight)
    if args.grid_min > 0:
        dev.set_option(N.RT_OPT_GRID_MIN, args.grid_min)
    if args.dirmask >= 0:
        dev.set_option(N.RT_OPT_DIRMASK, args.dirmask)
    if args.dirmask_k > 0:
        dev.set_option(N.RT_OPT_DIRMASK_K, args.dirmask_k)
    if args.dirmask_cells > 0:
        dev.set_option(N.RT_OPT_DIRMASK_CELLS, args.dirmask_cells)
    if args.dirlist >= 0:
        dev.set_option(N.RT_OPT_DIRLIST, args.dirlist)
    if args.dirlist_min:
        dev.set_option(N.RT_OPT_DIRLIST_MIN, args.dirlist_min)
    if args.dirlist_k:
        dev.set_option(N.RT_OPT_DIRLIST_K, args.dirlist_k)
    if args.dirlist_cells:
        dev.set_option(N.RT_OPT_DIRLIST_CELLS, args.dirlist_cells)
    if args.dirlist_reach:
        dev.set_option(N.RT_OPT_DIRLIST_REACH, args.dirlist_reach)
    if args.tail_spill >= 0:
        dev.set_option(N.RT_OPT_TAIL_SPILL, args.tail_spill)
    if args.tail_spill_single >= 0:
        dev.set_option(N.RT_OPT_TAIL_SPILL_SINGLE, args.tail_spill_single)
    target = RenderTarget.new(dev, (W, H))
    rt = RandomTexture(dev)
    rt.write(rnd)
    conn = Connection(scene, dev, sky, rt, max_objects=("live" if args.slots == 0 else args.slots))
    conn.update_buffers((W, H), scene, (0.0, 0.0))

    tiles = None
    gather_path = None
    if world == 1 and args.force_comm:
        from rsraytracingv3_amd.distributed import TileParallelRenderer
        gather_path = "rt_comm_init: one-rank communicator (rehearsal)"
        tiles = TileParallelRenderer(dev, W, H, args.strip_rows)
        if not tiles.preflight(conn, args.comm_timeout):
            raise SystemExit("bench.py: the one-rank pre-flight launch did not complete")
    if world > 1:
        from rsraytracingv3_amd.distributed import TileParallelRenderer, TorchTileRenderer
        # production: the gather inside librt355 over RCCL; a gloo rehearsal (ranks sharing one GPU, which RCCL
        # refuses) gathers through torch.distributed and host memory instead
        gather_path = "rt_comm_init: RCCL send/recv inside librt355"
        if args.backend == "nccl":
            from rsraytracingv3_amd.distributed import CommUnavailable
            try:
                # raises CommUnavailable on EVERY rank or on none (the ranks agree inside, before and after
                # ncclCommInitRank): no rank can be left waiting in a collective the others skipped
                tiles = TileParallelRenderer(dev, W, H, args.strip_rows, init_timeout_s=args.comm_timeout)
            except CommUnavailable as exc:  # reported, never silent: the line says which transport carried the gather
                gather_path = "torch.distributed.gather (TorchTileRenderer) — rt_comm_init unavailable: %s" % exc
                print("bench.py rank %d: %s" % (rank, gather_path), file=sys.stderr, flush=True)
                if exc.fatal:  # this rank's context is still inside ncclCommInitRank: nothing can be measured with it
                    sys.stderr.flush()
                    os._exit(3)
                tiles = TorchTileRenderer(dev, W, H, args.strip_rows)
            else:
                # pre-flight: one launch and its gather under a watchdog — an exchange that never completes must cost the
                # measurement its transport, not its life
                stuck = 0.0 if tiles.preflight(conn, args.comm_timeout) else 1.0
                flag = torch.tensor([stuck], device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                if flag.item() > 0.0:
                    tiles.abort()
                    gather_path = ("torch.distributed.gather (TorchTileRenderer) — the RCCL gather of the pre-flight launch did not "
                                   "complete within %.0f s on %s" % (args.comm_timeout, "this rank" if stuck else "another rank"))
                    print("bench.py rank %d: %s" % (rank, gather_path), file=sys.stderr, flush=True)
                    tiles = TorchTileRenderer(dev, W, H, args.strip_rows)
        else:
            gather_path = "torch.distributed.gather over %s, staged through host memory (rehearsal)" % args.backend
            tiles = TorchTileRenderer(dev, W, H, args.strip_rows)

    def run_frames(first, count, B):
        """frames first .. first+count-1 of the accumulation, B per launch; returns the number of launches"""
        k, launches = first, 0
        while k < first + count:
            n = min(B, first + count - k)
            j = jit[np.arange(k, k + n) % n_jit]
            if tiles is not None:  # rank 0 gets the previous launch's image assembled while this one renders
                tiles.render(conn, j[0] if B == 1 else j, k)
            elif B == 1:
                conn.frame_data.jitter = (float(j[0, 0]), float(j[0, 1]))
                conn.frame_data.progressive_count = k
                conn.draw()
            else:
                conn.draw_frames(j, k)
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

    def agree(flag):
        """Loop-exit decisions must be the same on every rank (each rank has its own clock)."""
        if world == 1:
            return flag
        t = torch.tensor([1.0 if flag else 0.0], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return bool(t.item() > 0.5)

    def timed_leg(B, steps, warm_steps, min_seconds, clock_warmup, frames_per_block=None):
        """Clock warm-up, `warm_steps` untimed steps, then blocks of exactly `steps` steps — a step = one launch of B
        frames — until min_seconds have been measured.  Returns the per-block records; the caller reports the median
        block.  `frames_per_block`: the round-2 protocol instead, a block = that many FRAMES, B per launch."""
        first = 0
        t_end = time.perf_counter() + clock_warmup
        while True:  # untimed: back-to-back launches so the clock has settled (DVFS) before anything is timed
            run_frames(first, 8 * B, B)
            first += 8 * B
            dev.sync()
            if not agree(time.perf_counter() < t_end):
                break
        block_frames = steps * B if frames_per_block is None else frames_per_block
        warm = warm_steps * B  # whole launches, so every launch a profiler sees has the timed shape
        if warm:
            run_frames(first, warm, B)
            first += warm
        fence()
        blocks, measured = [], 0.0
        while True:
            t0 = time.perf_counter()
            launches = run_frames(first, block_frames, B)
            fence()
            dt = time.perf_counter() - t0
            seg = np.zeros(launches, np.uint64)
            rough = np.zeros(launches, np.uint64)
            kms = np.zeros(launches, np.float32)
            rms = np.zeros(launches, np.float32)
            dev.call("rt_get_frame_history", launches, seg.ctypes.data_as(C.c_void_p), rough.ctypes.data_as(C.c_void_p),
                     kms.ctypes.data_as(C.c_void_p))
            dev.call("rt_get_resolve_history", launches, rms.ctypes.data_as(C.c_void_p))
            tms = np.zeros(launches, np.float32)
            dev.call("rt_get_tail_history", launches, tms.ctypes.data_as(C.c_void_p))
            span = C.c_float(0.0)
            dev.call("rt_get_launch_span", launches, C.byref(span))
            # device time of the block's render kernels: the sum of their HIP-event times — or, when launches overlap
            # (several in flight), the span from the first kernel's start to the last one's end, which counts the
            # device once
            blocks.append({"dt": dt, "launches": launches, "frames": block_frames, "seg": float(seg.sum()), "rough": float(rough.sum()),
                           "kernel_ms": min(float(kms.sum()), float(span.value)), "kernel_ms_sum": float(kms.sum()),
                           "span_ms": float(span.value), "resolve_ms": float(rms.sum()), "tail_ms": float(tms.sum()), "first": first,
                           "seg_first_launch": int(seg[0])})
            measured += dt
            if not agree(measured < min_seconds and len(blocks) < 4096):
                break
        return blocks

    def reduce_leg(blocks):
        """Median block (by the MAX-over-ranks time) and whole-job totals of that block."""
        st = dev.stats()
        frame_px = float(st.last_pixels) / max(int(st.last_batch), 1)  # this rank's pixels of one frame
        dts = torch.tensor([b["dt"] for b in blocks], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(dts, op=dist.ReduceOp.MAX)
        order = torch.argsort(dts)
        mid = int(order[(len(blocks) - 1) // 2])
        b = blocks[mid]
        tot = torch.tensor([b["seg"], b["rough"], frame_px], dtype=torch.float64, device="cuda")
        if world > 1:
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        return {"elapsed": float(dts[mid]), "block": b, "n_blocks": len(blocks), "total_seg": float(tot[0]),
                "total_rough": float(tot[1]), "total_px": float(tot[2]), "frame_px": frame_px, "stats": st,
                "dt_min": float(dts.min()), "dt_max": float(dts.max()), "measured_s": float(dts.sum())}

    # ---- headline leg: B frames per launch ------------------------------------------------------------------
    # frames per launch = frames per step: 8 per GPU of the job (64 is the most one launch takes)
    B = args.batch
    if B <= 0:
        B = min(64, 8 * world)
    head = reduce_leg(timed_leg(B, args.steps, args.warmup, args.min_seconds, args.clock_warmup))

    # ---- extra legs: one frame per launch (reference call pattern); 80 zero-padded slots ------------------
    def leg_summary(leg, frames_per_launch):
        blk = leg["block"]
        return {"frames_per_launch": frames_per_launch, "launches_in_flight": int(leg["stats"].last_launches_in_flight),
                "steps": args.steps, "frames_per_block": blk["frames"],
                "ms_per_step": round(leg["elapsed"] / args.steps * 1e3, 5),
                "mrays_per_s": round(leg["total_seg"] / leg["elapsed"] / 1e6, 2),
                "kernel_mrays_per_s": round(blk["seg"] / (blk["kernel_ms"] * 1e-3) / 1e6, 2) if blk["kernel_ms"] else None,
                "kernel": kernel_name(leg["stats"]), "blocks_timed": leg["n_blocks"]}

    single = slots80 = block_of_frames = None
    if not args.no_extra_legs:
        if B > 1:
            # the same frames as a headline block (steps x B of them), one frame per launch and no timing-only events (a render loop
            # does not ask for kernel times: RT_OPT_TIMING = 0)
            dev.set_option(N.RT_OPT_TIMING, 0)
            single = leg_summary(reduce_leg(timed_leg(1, args.steps, 8, args.min_seconds / 2, 0.25, frames_per_block=min(args.steps * B, 480))), 1)  # (the library keeps 512 launches of history)
            dev.set_option(N.RT_OPT_TIMING, 1)
            # the protocol of rounds 1-2: a block = --steps FRAMES (B per launch), i.e. a few launches between two barriers
            old = reduce_leg(timed_leg(B, args.steps, 2, args.min_seconds / 4, 0.1, frames_per_block=args.steps))
            block_of_frames = {"frames_per_block": args.steps, "launches_per_block": old["block"]["launches"],
                               "ms_per_block": round(old["elapsed"] * 1e3, 5),
                               "mrays_per_s": round(old["total_seg"] / old["elapsed"] / 1e6, 2)}
        if world == 1 and args.slots == 0 and len(scene.objects) <= 80:
            conn.max_objects = 80      # connection.rs: the fixed 80-slot buffer, zero-filled tail (scene.rs:28)
            conn.last_size = (0, 0)    # force the dirty branch of update_buffers
            conn.update_buffers((W, H), scene, (0.0, 0.0))
            leg80 = reduce_leg(timed_leg(B, args.steps, 8, args.min_seconds / 2, 0.25))
            slots80 = leg_summary(leg80, B)
            f80 = algorithmic_flops(leg80["frame_px"] * leg80["block"]["frames"], leg80["block"]["seg"], 80)
            slots80["valu_frac"] = round(f80 / (leg80["block"]["kernel_ms"] * 1e-3) / 1e12 / VALU_PEAK_TFLOPS, 5)
            conn.max_objects = "live"
            conn.last_size = (0, 0)
            conn.update_buffers((W, H), scene, (0.0, 0.0))
            dev.sync()

    # ---- N > 1: the assembled frame must be the single-GPU frame, bit for bit (per-pixel results do not depend on
    # the partition) — one frame through the communicator, then the same frame on rank 0 alone -------------------
    gather_check = None
    if tiles is not None and type(tiles).__name__ == "TileParallelRenderer":
        fence()
        probe = (0.125, -0.25)
        tiles.render(conn, probe, 0)
        assembled = tiles.finish(read=True)
        fence()
        tiles.close()  # every rank leaves the communicator; rank 0 now owns the whole frame
        tiles = None
        if rank == 0:
            try:
                conn.frame_data.jitter, conn.frame_data.progressive_count = probe, 0
                conn.draw()
                dev.sync()
                alone = target.read_rgba8()
                same = bool(alone.shape == assembled.shape and np.array_equal(alone, assembled))
                gather_check = {"assembled_frame_equals_single_gpu_frame": same,
                                "differing_texels": int((alone != assembled).any(-1).sum()) if alone.shape == assembled.shape else -1}
            except Exception as exc:  # the check must never cost the measurement
                gather_check = {"error": str(exc)}
        if world > 1:
            dist.barrier()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    steps = args.steps
    elapsed, blk = head["elapsed"], head["block"]
    n_frames = blk["frames"]  # of the median block: steps x B
    n_objects = len(scene.objects) if args.slots == 0 else args.slots
    value = head["total_seg"] / elapsed / 1e6
    # roofline of the dominant kernel (the render kernel), rank 0's launches of the median block
    launches = blk["launches"]
    k_sec = blk["kernel_ms"] * 1e-3
    b_alg = algorithmic_bytes(head["frame_px"] * n_frames, blk["seg"], blk["rough"])
    f_alg = algorithmic_flops(head["frame_px"] * n_frames, blk["seg"], n_objects)
    achieved = b_alg / k_sec / 1e9
    frac_alg = f_alg / k_sec / 1e12 / VALU_PEAK_TFLOPS
    hst = head["stats"]
    structure = ("candidate lists, RT_OPT_DIRLIST" if hst.dirmask_lists else "direction-mask table, RT_OPT_DIRMASK") if hst.dirmask_cells else (
        "sphere grid, RT_OPT_GRID_MIN" if hst.grid_cells else None)
    # Measured HBM traffic and issue utilisation are IMPORTED from the committed PMC passes (separate rocprofv3 --pmc
    # runs cannot happen inside this process): quoted only when the profile was collected on THIS workload, this launch
    # shape AND these kernel sources (sha256 over csrc/ + include/, recorded by tools/pmc_summary.py); null otherwise.
    traffic = traffic_resolve = issue_util = traffic_source = None
    workload_key = {"width": W, "height": H, "objects": n_objects, "bounces": args.bounces, "sky_width": args.sky, "n_gpus": world}
    source_now = N.kernel_source_hash()
    for name in ("traffic_r04.json", "traffic_r04_config3.json", "traffic_r03.json", "traffic_r02.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(tpath) or traffic is not None:
            continue
        try:
            with open(tpath) as f:
                prof = json.load(f)
        except Exception:
            continue
        prof_key = dict(prof.get("workload_key", {}))
        prof_key.pop("frames_per_launch", None)  # the profile's launches carried 8 frames; traffic is taken per frame
        tuned = (args.variant != -1 or args.dirmask >= 0 or args.dirmask_k or args.dirmask_cells or args.grid_min or args.flags or
                 args.dirlist >= 0 or args.dirlist_min or args.dirlist_k or args.dirlist_cells or args.dirlist_reach or
                 args.batch_inflight != 0 or args.blocks_per_cu or args.refill_min != 1 or args.tile_chunk != 2 or args.tail_spill >= 0 or args.tail_spill_single >= 0)
        if prof_key != workload_key or B <= 1 or tuned:  # the profile was collected with the library's defaults
            continue
        if prof.get("kernel_source_sha256") != source_now:
            traffic_source = {"file": "profiles/" + name, "used": False,
                              "why": "collected on other kernel sources (sha256 %s..., this tree %s...)"
                                     % (str(prof.get("kernel_source_sha256"))[:12], source_now[:12])}
            continue
        # per launch of THIS run: the profile's bytes per frame x the frames an average launch of the block carried
        frames_per_launch = float(n_frames) / launches
        per_frame = prof.get("render_kernel", {}).get("hbm_bytes_per_frame")
        traffic = round(per_frame * frames_per_launch, 1) if per_frame else None
        per_frame = prof.get("resolve_kernel", {}).get("hbm_bytes_per_frame")
        traffic_resolve = round(per_frame * frames_per_launch, 1) if per_frame else None
        issue_util = prof.get("render_kernel", {}).get("valu_issue_util")
        traffic_source = {"file": "profiles/" + name, "used": True, "kernel_source_sha256": source_now[:16],
                          "how": "imported, not measured in this run: separate rocprofv3 --pmc passes of this workload on these "
                                 "kernel sources (tools/profile_round.sh), bytes per frame x frames per launch of this run"}
    out = {
        "metric": "Mrays/s at 1920x1080, 64 spheres, 8 bounces",
        "value": round(value, 2),
        "unit": "Mrays/s",
        "n_gpus": world,
        "steps": steps,
        "warmup": args.warmup,
        "ms_per_step": round(elapsed / steps * 1e3, 5),
        "higher_is_better": True,
        "scaling": "weak",  # a step is 8 x N frames, each split over the N GPUs: eight frames' worth of pixels per GPU and step
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "%dx%d, %d spheres + ground plane (%d object slots), %d bounces, 1 spp, %dx%d RGBA32F skybox"
                        % (W, H, args.spheres, n_objects, args.bounces, args.sky, args.sky // 2),
            "partition": "row strips of %d, round-robin over %d GPU(s)%s" % (
                args.strip_rows, world, ", strips gathered on rank 0 after every launch, beside the next one" if world > 1 else ""),
            "gather": gather_path,
            "gather_check": gather_check,
            "step": "one launch of %d consecutive frames of the accumulation (rt_render_frames), every frame tile-split over the %d "
                    "GPU(s): %d frames' worth of pixels per GPU and step, one gather per step" % (B, world, B // world),
            "frames_per_step": B,
            "headline_mode": "%d consecutive frames of the accumulation per launch (%d per GPU), one gather per launch; the "
                             "one_frame_per_launch leg is the per-frame figure (one launch and one gather per frame)" % (B, B // world),
            "kernel_variant": args.variant,
            "frames_per_launch": B,
            "progressive_count": "frame index k (1-spp frames of a resting camera accumulate, app.rs:208-254)",
            "timing": "median of %d blocks of %d steps = %d frames (%.2f s measured, block %.3f..%.3f ms) after %.1f s of untimed "
                      "launches + %d warm-up steps" % (head["n_blocks"], steps, n_frames, head["measured_s"], head["dt_min"] * 1e3,
                                                      head["dt_max"] * 1e3, args.clock_warmup, args.warmup),
            "one_frame_per_launch": single,
            "reference_80_slots": slots80,
            "block_of_frames": block_of_frames,
            "segments_per_pixel": round(head["total_seg"] / (float(W) * H * n_frames), 4),
            "mpixels_per_s": round(W * H * n_frames / elapsed / 1e6, 2),
        },
        # The binding roofline of this path is the FP32 vector ALU (77 flop per algorithmic byte against a machine
        # balance of ~20), so `roofline` is the VALU one; the HBM figures sit in roofline.hbm.
        "roofline": {
            "bound": "valu",
            "achieved": round(f_alg / k_sec / 1e12, 3),
            "peak": VALU_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            # achieved / peak in units of the REFERENCE's algorithm (below): printed as `frac` only while it is a fraction
            "frac": round(frac_alg, 5) if frac_alg <= 1.0 else None,
            "frac_alg_reference_units_may_exceed_1": round(frac_alg, 5),
            "issue_util": issue_util,  # the hardware view: SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x busy cycles), imported with `traffic`
            "achieved_basis": "reference-equivalent flops: SURVEY.md 8d's F_alg = S (25 N + 200) + 60 P charges the reference's linear scan of "
                              "all N objects (ray.hlsl:97-107) to every one of the S segments; / HIP-event kernel time.  %s"
                              % ("These launches looked their candidates up (%s) and tested a few of the N, so this is speed in units of the "
                                 "reference's algorithm, not ALU utilisation — it can exceed the peak; `issue_util` is the hardware's fraction."
                                 % structure if structure else "These launches scanned every sphere for every secondary segment, as the reference does."),
            "traffic": traffic,
            "traffic_source": traffic_source,
            "kernel": kernel_name(head["stats"]),
            "avg_launch_ms": round(blk["kernel_ms"] / launches, 5),
            "launches_in_flight": int(head["stats"].last_launches_in_flight),
            "kernel_avg_duration_ms": round(blk["kernel_ms_sum"] / launches, 5),
            "timing_note": ("launches overlap (RT_OPT_BATCH_INFLIGHT = %d): avg_launch_ms = HIP-event span from the first kernel's start "
                            "to the last kernel's end of the block / launches (device time counted once); kernel_avg_duration_ms = "
                            "mean HIP-event duration of one kernel, which is what a kernel trace averages and includes the time "
                            "it shared the device with its neighbour; --batch-inflight 1 gives non-overlapping kernels"
                            % int(head["stats"].last_launches_in_flight)) if (B > 1 and int(head["stats"].last_launches_in_flight) > 1) else
                           ("one launch at a time: avg_launch_ms is the mean HIP-event duration of the launch's render step — "
                            "render_blocks followed by tail_stream, which finishes the paths the workgroups hand over (RT_OPT_TAIL_SPILL); "
                            "tail_avg_launch_ms is tail_stream's share of it"
                            if blk.get("tail_ms") else
                            "one launch at a time: avg_launch_ms is the mean HIP-event duration of the kernel"),
            "tail_avg_launch_ms": round(blk["tail_ms"] / launches, 5) if blk.get("tail_ms") else None,
            "resolve_avg_launch_ms": round(blk["resolve_ms"] / launches, 5),
            "launches": launches,
            "alg_flops_per_launch": round(f_alg / launches, 1),
            "kernel_mrays_per_s": round(blk["seg"] / k_sec / 1e6, 2),
            "scan_lane_occupancy": round(float(head["stats"].last_segments) / (64.0 * head["stats"].last_wave_iterations), 4)
            if head["stats"].last_wave_iterations else None,
            "alg_bytes_per_launch": round(b_alg / launches, 1),
            "hbm": {
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 5),
                "achieved_basis": "algorithmic bytes (SURVEY.md 8d: 8 B/pixel + 16 B/segment + 8 B/rough hit) / HIP-event kernel time",
                "alg_bytes_per_launch": round(b_alg / launches, 1),
                "traffic": traffic,
                "traffic_resolve_kernel": traffic_resolve,
                "measured_gbs": round(traffic / (k_sec / launches) / 1e9, 2) if traffic else None,
            },
        },
    }
    # (kept for the tools that read it: the same figures as `roofline`, which is the VALU roofline now)
    out["valu"] = {k: out["roofline"][k] for k in ("achieved", "peak", "unit", "frac", "frac_alg_reference_units_may_exceed_1", "issue_util",
                                                   "alg_flops_per_launch", "kernel_mrays_per_s", "scan_lane_occupancy", "achieved_basis")}
    if world == 1 and not args.no_cpu_baseline:
        import oracle
        rows = (0, H) if args.cpu_rows <= 0 else ((H - args.cpu_rows) // 2, (H - args.cpu_rows) // 2 + args.cpu_rows)
        if args.slots == 0:
            objects, count = scene.live_object_bytes(), len(scene.objects)
        else:
            objects, count = scene.as_bytes(W, H, args.slots)[0], args.slots
        # every host core this process may use (all cores, count stated): the CPUs of its affinity mask, capped by the
        # cgroup's CPU quota where there is one (a GPU box gives one GPU a share of its host's cores)
        threads = max(1, min(oracle.max_threads(), usable_cores()))
        cfg = scene.config_bytes(W, H)
        k0 = blk["first"]  # first frame of the median block: its first launch covers frames k0 .. k0+first_n-1
        oracle.render(objects, count, cfg, jit[k0 % n_jit], 0, sky, rnd, rows=(rows[0], rows[0] + 8),
                      threads=threads, want_hits=False)  # warm the thread pool
        cseg, cdt, frames, first_seg = 0, 0.0, 0, 0
        first_n = B
        while (cdt < 12.0 or frames < first_n) and frames < 512:  # bounded sample: ~12 s of whole frames
            k = k0 + frames
            c0 = time.perf_counter()
            fr = oracle.render(objects, count, cfg, jit[k % n_jit], 0, sky, rnd, rows=rows, threads=threads, want_hits=False)
            cdt += time.perf_counter() - c0
            fs = int(fr.segments[rows[0]:rows[1]].sum())
            first_seg += fs if frames < first_n else 0
            cseg += fs
            frames += 1
        out["cpu_baseline"] = {
            "value": round(cseg / cdt / 1e6, 3),
            "unit": "Mrays/s",
            "cores": threads,
            "host_cores": os.cpu_count(),
            "kind": "port",
            "sample": "%d frame(s), rows %d..%d of the same %dx%d workload (%d segments, %.1f s wall, %d OpenMP "
                      "threads = every core this process may use), CPU oracle oracle/rt_oracle.c built -O3 -fopenmp; the reference itself cannot run here (no "
                      "Rust/wgpu/Vulkan)" % (frames, rows[0], rows[1], W, H, cseg, cdt, threads),
        }
        # cross-check: the GPU counted the same number of segments for the frames of that launch
        out["cpu_baseline"]["segments_match_gpu"] = bool(args.cpu_rows <= 0 and first_seg == blk["seg_first_launch"])
    sys.stdout.flush()
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
