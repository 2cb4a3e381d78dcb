"""Tile-parallel frames over the GPUs of one node: row strips per rank + one gather.

One process per GPU (``torch.distributed``; backend ``nccl`` is RCCL over xGMI on ROCm).
The render path itself has no cross-GPU dependency — strips are independent and the
accumulation history of a pixel stays on the rank that owns it — so the
only collective is the gather that assembles the image on rank 0: every peer sends its
compact strip buffer straight to the root (RCCL lowers ``gather`` to grouped
send/recv, one xGMI link per peer), never a ring.  PyTorch is used for device memory,
streams and the process group only.

``TileParallelRenderer`` is the production path: the gather runs INSIDE ``librt355.so`` (``rt_comm_init``:
grouped ncclSend / ncclRecv to rank 0 on a communication stream of the context, double-buffered targets, strips
interleaved by a kernel on rank 0) — ``torch.distributed`` only carries the 128-byte communicator id to the other
ranks.  ``FrameGatherer`` / ``TorchTileRenderer`` do the same composition with ``torch.distributed`` collectives
(gloo/CPU in the unit tests, or a rehearsal with several ranks on one GPU).
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from . import partition


class FrameGatherer:
    """Gathers per-rank compact strip buffers (padded to the largest piece) on rank 0 and
    de-interleaves them into the (height, width[, C]) frame with one index_select."""

    def __init__(self, height: int, width: int, strip_rows: int, device: torch.device, group=None,
                 force_collective: bool = False):
        self.group = group
        self.force_collective = force_collective  # world 1 still goes through dist.gather (RCCL smoke test)
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.height, self.width, self.strip_rows = height, width, strip_rows
        self.device = device
        self.local_rows = partition.local_row_count(height, self.rank, self.world, strip_rows)
        self.padded_rows = partition.max_local_rows(height, self.world, strip_rows)
        # frame row y lives at row rank(y)*padded_rows + local(y) of the stacked gather buffer
        src = np.empty(height, np.int64)
        for r in range(self.world):
            idx = partition.local_row_indices(height, r, self.world, strip_rows)
            src[idx] = r * self.padded_rows + np.arange(idx.size)
        self.row_source = torch.from_numpy(src).to(device) if self.rank == 0 else None
        self._stack = {}
        # rehearsal mode: a gloo group with device tensors stages the gather through host memory
        self.via_host = (dist.is_initialized() and dist.get_backend(group) == "gloo" and device.type == "cuda")

    def new_local(self, dtype=torch.int32, channels: Optional[int] = None) -> torch.Tensor:
        """A send buffer of padded_rows x width [x channels]; the render kernel fills the first
        local_rows rows."""
        shape = (self.padded_rows, self.width) + ((channels,) if channels else ())
        return torch.zeros(shape, dtype=dtype, device=self.device)

    def gather(self, local: torch.Tensor, async_op: bool = False):
        """Collective.  Returns (stacked buffer on rank 0 | None, work handle | None)."""
        if self.world == 1 and not (self.force_collective and dist.is_initialized()):
            return local.unsqueeze(0), None
        if self.via_host:
            torch.cuda.current_stream(self.device).synchronize()
            host = local.cpu()
            parts = [torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
            dist.gather(host, parts, dst=0, group=self.group)
            return (torch.stack(parts).to(self.device) if self.rank == 0 else None), None
        key = (local.dtype, tuple(local.shape))
        stack = None
        if self.rank == 0:
            stack = self._stack.get(key)
            if stack is None:
                stack = torch.empty((self.world,) + tuple(local.shape), dtype=local.dtype, device=self.device)
                self._stack[key] = stack
        work = dist.gather(local, list(stack.unbind(0)) if self.rank == 0 else None, dst=0, group=self.group,
                           async_op=async_op)
        return stack, work

    def assemble(self, stacked: torch.Tensor) -> torch.Tensor:
        """Rank 0: stacked (world, padded_rows, W[, C]) -> frame (height, W[, C])."""
        flat = stacked.reshape((-1,) + tuple(stacked.shape[2:]))
        return flat.index_select(0, self.row_source)


class TorchTileRenderer:
    """The gather through ``torch.distributed`` instead of the library's own RCCL path — for rehearsals where RCCL
    cannot run (several ranks sharing one GPU: gloo, staged through host memory) and as a second, independent
    composition to test against.  rt355 ``Device`` + ``FrameGatherer``: renders this rank's strips on the current torch
    stream into double-buffered send tensors and gathers the sRGB8 texels on rank 0.

    Pipeline per ``render`` call k (so that gather(k-1) runs beside render(k)):
      1. bind send buffer k%2 (free: gather(k-2) was waited for during call k-1), launch render(k);
      2. wait for gather(k-1) and de-interleave frame k-1 on rank 0 (returned to the caller);
      3. start gather(k) asynchronously — RCCL orders it after render(k) on the current stream.
    ``finish`` drains the pipeline and returns the last frame."""

    def __init__(self, rt_device, width: int, height: int, strip_rows: int = 8, group=None,
                 force_collective: bool = False):
        self.rt = rt_device
        self.torch_device = torch.device("cuda", rt_device.device_id)
        self.gatherer = FrameGatherer(height, width, strip_rows, self.torch_device, group, force_collective)
        g = self.gatherer
        rt_device.set_partition(g.rank, g.world, strip_rows)
        self.stream = torch.cuda.current_stream(self.torch_device)
        rt_device.call("rt_set_stream", self.stream.cuda_stream)
        self.buffers = [g.new_local(torch.int32), g.new_local(torch.int32)]
        self._ptrs = [b.data_ptr() for b in self.buffers]
        self._bind = rt_device._lib.rt_bind_outputs   # hot path: raw entry points, status checked inline
        self._render = rt_device._lib.rt_render
        self._render_frames = rt_device._lib.rt_render_frames
        self.frame_index = 0
        self.pending = None  # (work handle | None, stacked buffer | None) of the gather in flight

    def _collect(self):
        """Wait for the gather in flight; rank 0 gets the assembled (height, width) int32 frame."""
        if self.pending is None:
            return None
        work, stacked = self.pending
        self.pending = None
        if work is not None:
            work.wait()
        return self.gatherer.assemble(stacked) if self.gatherer.rank == 0 else None

    def render(self, connection, jitter, progressive_count: int = 0):
        """Launch frame k; returns frame k-1 (rank 0) or None.  ``jitter`` is one (x, y) pair, or a
        sequence of n pairs: then n consecutive frames of the accumulation (counts progressive_count,
        +1, ...) are rendered in ONE launch (rt_render_frames) and gathered once, after the last —
        a rank's strips of a single frame are too little work to fill a GPU."""
        k = self.frame_index & 1
        buf = self.buffers[k]
        rc = self._bind(self.rt.handle, self._ptrs[k], self._ptrs[k ^ 1], None)
        jit = np.asarray(jitter, dtype=np.float32)
        if jit.ndim == 1:
            if rc == 0:
                rc = self._render(self.rt.handle, float(jit[0]), float(jit[1]), progressive_count)
            last = (float(jit[0]), float(jit[1]), progressive_count)
        else:
            n = jit.shape[0]
            blob = np.zeros((n, 4), dtype=np.uint32)  # FrameData: jitter f32 x2, progressive_count u32, pad
            blob[:, :2] = jit.view(np.uint32)
            blob[:, 2] = np.arange(progressive_count, progressive_count + n, dtype=np.uint32)
            if rc == 0:
                rc = self._render_frames(self.rt.handle, n, blob.ctypes.data)
            last = (float(jit[-1, 0]), float(jit[-1, 1]), progressive_count + n - 1)
        self.rt._check(rc, "rt_render")
        connection.frame_data.jitter = last[:2]
        connection.frame_data.progressive_count = last[2]
        self.frame_index += 1
        previous_frame = self._collect()
        stacked, work = self.gatherer.gather(buf, async_op=True)
        self.pending = (work, stacked)
        return previous_frame

    def finish(self):
        """Drain: returns the last frame on rank 0 (None elsewhere) after a stream sync."""
        frame = self._collect()
        torch.cuda.current_stream(self.torch_device).synchronize()
        return frame


class CommUnavailable(RuntimeError):
    """The RCCL communicator of ``TileParallelRenderer`` could not be formed.  Raised on EVERY rank of the group (the
    ranks agree before any of them raises), so the caller can switch all of them to another transport.  ``fatal``:
    this rank is still inside ``ncclCommInitRank`` on a helper thread (a peer never arrived) — its context must not
    be used again; exit the process."""

    def __init__(self, text: str, fatal: bool = False):
        super().__init__(text)
        self.fatal = fatal


def _all_ranks(ok: bool, group=None) -> bool:
    """Collective on the process group: True only when every rank says True."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return bool(ok)
    on_gpu = dist.get_backend(group) == "nccl"
    t = torch.tensor([1.0 if ok else 0.0], device=torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return bool(t.item() > 0.5)


class TileParallelRenderer:
    """One process per GPU, the image gather inside the C-ABI library (``rt_comm_init``, ``include/rt355.h``).

    Every rank issues the same ``render`` calls; after each launch the rank's strips travel to rank 0 over RCCL
    beside the next launch's rendering.  ``finish`` waits and returns the assembled (height, width, 4) uint8 frame
    on rank 0.  The process group (any backend) only carries the hand-shake: whether every rank can bind RCCL, rank
    0's communicator id, and whether every rank joined.

    Forming the communicator cannot leave the ranks in different states: (1) every rank probes RCCL without a
    collective (``rt_comm_available``) and the ranks agree before anyone enters ``ncclCommInitRank``; (2) rank 0
    always broadcasts an (ok, id) pair, also when it could not make an id; (3) ``rt_comm_init`` runs on a helper
    thread under ``init_timeout_s`` and the ranks agree on its outcome.  Any failure raises ``CommUnavailable`` on
    every rank."""

    def __init__(self, rt_device, width: int, height: int, strip_rows: int = 8, group=None, init_timeout_s: float = 120.0):
        import threading
        self.rt = rt_device
        self.width, self.height = width, height
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        cls = type(rt_device)
        # (1) can every rank bind RCCL at all?
        try:
            usable, text = cls.comm_available()
        except Exception as exc:  # e.g. a library without the entry point
            usable, text = False, str(exc)
        if not _all_ranks(usable, group):
            raise CommUnavailable("no RCCL on rank %d: %s" % (self.rank, text) if not usable else "another rank cannot bind RCCL")
        # (2) rank 0's id, or why there is none — a broadcast either way
        ident = [None]
        if self.rank == 0:
            try:
                ident = [(True, cls.comm_unique_id())]
            except Exception as exc:
                ident = [(False, str(exc))]
        if self.world > 1:
            dist.broadcast_object_list(ident, src=0, group=group)
        if not ident[0][0]:
            raise CommUnavailable("rank 0 could not create the communicator id: %s" % ident[0][1])
        # (3) the collective itself, bounded
        outcome = {}

        def join():
            try:
                rt_device.comm_init(ident[0][1], self.rank, self.world, strip_rows)
                outcome["ok"] = True
            except Exception as exc:
                outcome["error"] = str(exc)

        worker = threading.Thread(target=join, name="rt_comm_init", daemon=True)
        worker.start()
        worker.join(init_timeout_s)
        stuck = worker.is_alive()
        joined = bool(outcome.get("ok")) and not stuck
        if not _all_ranks(joined, group):
            if joined:
                rt_device.comm_abort()  # the peers are not there: never wait for them
            why = ("rt_comm_init still running after %.0f s on rank %d" % (init_timeout_s, self.rank)) if stuck else \
                  (("rt_comm_init failed on rank %d: %s" % (self.rank, outcome.get("error"))) if not joined
                   else "rt_comm_init failed on another rank")
            raise CommUnavailable(why, fatal=stuck)
        self._render = rt_device._lib.rt_render
        self._render_frames = rt_device._lib.rt_render_frames

    def render(self, connection, jitter, progressive_count: int = 0):
        """Launch one frame, or — ``jitter`` a sequence of n pairs — n consecutive frames of the accumulation in one
        launch (counts progressive_count, +1, ...); the gather of this launch overlaps the next one."""
        jit = np.asarray(jitter, dtype=np.float32)
        if jit.ndim == 1:
            rc = self._render(self.rt.handle, float(jit[0]), float(jit[1]), progressive_count)
            last = (float(jit[0]), float(jit[1]), progressive_count)
        else:
            n = jit.shape[0]
            blob = np.zeros((n, 4), dtype=np.uint32)  # FrameData: jitter f32 x2, progressive_count u32, pad
            blob[:, :2] = jit.view(np.uint32)
            blob[:, 2] = np.arange(progressive_count, progressive_count + n, dtype=np.uint32)
            rc = self._render_frames(self.rt.handle, n, blob.ctypes.data)
            last = (float(jit[-1, 0]), float(jit[-1, 1]), progressive_count + n - 1)
        self.rt._check(rc, "rt_render")
        connection.frame_data.jitter = last[:2]
        connection.frame_data.progressive_count = last[2]
        return None

    def finish(self, read: bool = False):
        """Wait for rendering and the last gather; with ``read`` rank 0 gets the assembled frame."""
        self.rt.sync()
        if read and self.rank == 0:
            return self.rt.read_frame_rgba8(self.width, self.height)
        return None

    def preflight(self, connection, timeout_s: float = 90.0) -> bool:
        """One launch and its gather, polled instead of waited for: False when the exchange has not completed after
        ``timeout_s`` (a peer that never posts its side would otherwise hang every later call).  Every rank must call
        it; the caller agrees on the outcome across ranks and, on failure, calls ``abort`` everywhere."""
        import time
        self.render(connection, (0.0, 0.0), 0)
        deadline = time.monotonic() + timeout_s
        while not self.rt.comm_idle():
            if time.monotonic() > deadline:
                return False
            time.sleep(0.005)
        return True

    def abort(self):
        self.rt.comm_abort()

    def close(self):
        self.rt.comm_destroy()
