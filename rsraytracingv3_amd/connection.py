"""Host feeder of the render path — mirror of the reference's ``src/gpu/{connection,
render_target,random_texture}.rs`` on top of the rt355 C-ABI.

Same names and argument meaning as the reference where a counterpart exists:

=====================================  ==========================================
reference                              here
=====================================  ==========================================
``Initial::gpu_setup`` (app.rs:64-104)  ``Device(device_id)`` — owns the rt_context
``FrameData`` (connection.rs:13-40)     ``FrameData``
``RenderTarget`` (render_target.rs)     ``RenderTarget.new / resize / update``
``RandomTexture::start`` (:20-64)       ``RandomTexture.start`` (optional 4 ms thread)
``Connection::new`` (:297-368)          ``Connection(scene, device, ...)``
``Connection::update_buffers`` (:370)   ``Connection.update_buffers(size, scene)``
``App::render_scene`` (app.rs:208-254)  ``render_scene(device, connection, target, scene)``
=====================================  ==========================================

Differences forced by determinism: jitter and the random texture are
explicit inputs (seeded generators by default) instead of unseeded process-global RNGs.
"""
from __future__ import annotations

import copy
import os
import ctypes as C
import struct
import threading
import time
from typing import Optional, Tuple

import numpy as np

from . import _native as N
from . import synth
from .bytes import bytes_concat, f32_bytes, u32_bytes
from .rand_compat import fma_f32
from .scene import Scene


class Device:
    """One gfx950 GPU bound through ``rt_create`` (app.rs:64-104 binds one wgpu adapter)."""

    def __init__(self, device_id: int = 0):
        self._lib = N.load()
        handle = C.c_void_p()
        rc = self._lib.rt_create(device_id, C.byref(handle))
        if rc != N.RT_OK:
            raise N.RtError(rc, "rt_create(device %d)" % device_id,
                            "no usable gfx950 device — rt355 has no CPU fallback")
        self.handle = handle
        self.device_id = device_id

    def _check(self, rc: int, where: str):
        if rc != N.RT_OK:
            raise N.RtError(rc, where, (self._lib.rt_last_error(self.handle) or b"").decode())

    def call(self, name: str, *args):
        self._check(getattr(self._lib, name)(self.handle, *args), name)

    def close(self):
        if self.handle:
            self._lib.rt_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- thin wrappers ------------------------------------------------------
    def set_partition(self, rank: int, world: int, strip_rows: int = 8):
        self.call("rt_set_partition", rank, world, strip_rows)

    def set_outputs(self, flags: int):
        self.call("rt_set_outputs", flags)

    def set_accum_mode(self, mode: int):
        self.call("rt_set_accum_mode", mode)

    def set_variant(self, variant: int):
        self.call("rt_set_variant", variant)

    def set_option(self, option: int, value: int):
        self.call("rt_set_option", option, value)

    def sync(self):
        self.call("rt_sync")

    def stats(self) -> N.RtStats:
        st = N.RtStats()
        self.call("rt_get_stats", C.byref(st))
        return st

    # -- multi-GPU frame (one process per GPU; the gather runs inside the library over RCCL) ---------------
    @staticmethod
    def comm_available():
        """rt_comm_available: (True, path of the RCCL bound) or (False, why not).  Not a collective; needs no GPU."""
        buf = C.create_string_buffer(1024)
        rc = N.load().rt_comm_available(buf, len(buf))
        return rc == N.RT_OK, buf.value.decode("utf-8", "replace")

    @staticmethod
    def comm_unique_id() -> bytes:
        """Rank 0: the 128 opaque bytes every rank passes to ``comm_init`` (hand them over by any channel)."""
        lib = N.load()
        buf = C.create_string_buffer(128)
        rc = lib.rt_comm_get_unique_id(buf)
        if rc != N.RT_OK:
            raise N.RtError(rc, "rt_comm_get_unique_id", "no RCCL library could be loaded")
        return buf.raw

    def comm_init(self, unique_id: bytes, rank: int, world: int, strip_rows: int = 8):
        """Collective over all ranks: joins the communicator and takes the row-strip partition."""
        if len(unique_id) != 128:
            raise ValueError("the communicator id is 128 bytes")
        self.call("rt_comm_init", C.c_char_p(unique_id), rank, world, strip_rows)

    def comm_destroy(self):
        self.call("rt_comm_destroy")

    def comm_idle(self) -> bool:
        """rt_comm_idle: has everything queued on the context (renders, resolves, gathers) completed?  Never waits."""
        idle = C.c_int(0)
        self.call("rt_comm_idle", C.byref(idle))
        return bool(idle.value)

    def comm_abort(self):
        """rt_comm_abort: give up on an exchange that does not complete; the context is single-GPU again."""
        self.call("rt_comm_abort")

    def read_frame_rgba8(self, width: int, height: int) -> np.ndarray:
        """The assembled frame (rank 0 of a communicator, or the only GPU)."""
        out = np.empty((height, width, 4), np.uint8)
        self.call("rt_read_frame_rgba8", out.ctypes.data_as(C.c_void_p), out.nbytes)
        return out

    def device_buffers(self) -> N.RtDeviceBuffers:
        b = N.RtDeviceBuffers()
        self.call("rt_get_device_buffers", C.byref(b))
        return b


class _MultiLib:
    """``rt_x(handle, ...)`` -> ``rt_multi_x(handle, ...)``, so that Connection / RenderTarget / RandomTexture drive an
    ``rt_multi`` with the calls they make on one device.  ``rt_read_rgba8`` is the assembled frame (all rows are local to
    the rt_multi as a whole); calls without a multi-device form raise."""
    _NO_HANDLE = ("rt_exr_last_error", "rt_exr_read_rgba32f", "rt_exr_free", "rt_abi_version", "rt_has_ab_shapes")

    def __init__(self, lib):
        self._real = lib

    def __getattr__(self, name):  # called once per name: the resolved entry point is kept on the instance
        if name in self._NO_HANDLE or name.startswith("rt_multi_"):
            fn = getattr(self._real, name)
        elif name == "rt_read_rgba8":
            fn = self._real.rt_multi_read_frame_rgba8
        elif name.startswith("rt_") and hasattr(self._real, "rt_multi_" + name[3:]):
            fn = getattr(self._real, "rt_multi_" + name[3:])
        else:
            raise AttributeError("%s has no multi-device form (include/rt355.h: rt_multi_*)" % name)
        setattr(self, name, fn)
        return fn


class MultiDevice(Device):
    """Several gfx950 GPUs behind ONE process (``rt_multi_create``): the reference's shape — one process, one loop
    (app.rs:64-104, 208-254) — with the frame's row strips dealt over the devices and gathered on the first one by peer
    copies.  Drop-in for ``Device`` wherever the whole frame is meant: ``Connection``, ``RenderTarget``,
    ``RandomTexture`` and ``render_scene`` work unchanged; ``read_frame_rgba8`` / ``RenderTarget.read_rgba8`` return the
    assembled frame.  ``device_ids`` may repeat a device (tests on one GPU)."""

    def __init__(self, device_ids, strip_rows: int = 8):
        real = N.load()
        self._lib = _MultiLib(real)
        ids = (C.c_int * len(device_ids))(*device_ids)
        handle = C.c_void_p()
        rc = real.rt_multi_create(ids, len(device_ids), strip_rows, C.byref(handle))
        if rc != N.RT_OK:
            raise N.RtError(rc, "rt_multi_create(devices %s)" % list(device_ids),
                            "no usable gfx950 device — rt355 has no CPU fallback")
        self.handle = handle
        self.device_id = device_ids[0]
        self.device_ids = list(device_ids)

    def close(self):
        if self.handle:
            self._lib.rt_multi_destroy(self.handle)
            self.handle = C.c_void_p()

    def stats(self) -> N.RtStats:
        st = N.RtStats()
        self.call("rt_get_stats", C.byref(st))
        return st

    def member(self, index: int) -> C.c_void_p:
        """Device ``index``'s rt_context, for per-device calls on the raw library (rt_get_stats, rt_last_error)."""
        return C.c_void_p(self._lib.rt_multi_context(self.handle, index))

    def set_outputs(self, flags: int):
        if flags != N.RT_OUT_RGBA8:
            raise ValueError("an rt_multi renders the sRGB8 target only")

    def set_partition(self, rank: int, world: int, strip_rows: int = 8):
        raise ValueError("the partition belongs to the rt_multi")

    def read_frame_rgba8(self, width: int, height: int) -> np.ndarray:
        out = np.empty((height, width, 4), np.uint8)
        self.call("rt_read_frame_rgba8", out.ctypes.data_as(C.c_void_p), out.nbytes)
        return out


class FrameData:
    """connection.rs:13-40."""
    BUFFER_SIZE = 16
    JITTER_STRENGTH = 0.99

    def __init__(self, jitter: Tuple[float, float] = (0.0, 0.0)):
        self.jitter = (float(jitter[0]), float(jitter[1]))
        self.progressive_count = 0

    def as_bytes(self) -> bytes:
        """connection.rs:30-40."""
        return bytes_concat(self.BUFFER_SIZE, [f32_bytes(*self.jitter), u32_bytes(self.progressive_count)])


class RenderTarget:
    """render_target.rs:1-76 — the Rgba8UnormSrgb target; here also the float output."""

    def __init__(self, device: Device, initial_size: Tuple[int, int]):
        self.device = device
        self.size = tuple(initial_size)
        self.id = None
        device.call("rt_resize", self.size[0], self.size[1])

    @staticmethod
    def new(device: Device, initial_size: Tuple[int, int]) -> "RenderTarget":
        """render_target.rs:39-48."""
        return RenderTarget(device, initial_size)

    def resize(self, device: Device, size: Tuple[int, int]):
        """render_target.rs:72-75 (history is resized too — fixes SURVEY.md §0.7)."""
        self.size = tuple(size)
        device.call("rt_resize", self.size[0], self.size[1])

    def update(self) -> np.ndarray:
        """render_target.rs:50-70 hands the texture view to egui; here: read the texels back."""
        return self.read_rgba8()

    def _local_rows(self) -> int:
        return self.device.stats().local_rows

    def present_async(self, slot: int = 0):
        """render_target.rs:50-70 without stalling the render loop: queue the copy of the frames launched so far into pinned
        staging buffer ``slot``; it runs beside the next frames' rendering (``rt_present_async``)."""
        self.device.call("rt_present_async", slot)

    def present_wait(self, slot: int = 0, copy: bool = True) -> np.ndarray:
        """The texels queued by ``present_async(slot)`` (blocks until they have arrived).  ``copy=False`` returns a view of the
        pinned buffer, valid until the slot's next ``present_async``."""
        ptr, size, rows, width = C.c_void_p(), C.c_size_t(), C.c_uint32(), C.c_uint32()
        self.device.call("rt_present_wait", slot, C.byref(ptr), C.byref(size), C.byref(rows), C.byref(width))
        buf = (C.c_uint8 * size.value).from_address(ptr.value)
        view = np.frombuffer(buf, np.uint8).reshape(rows.value, width.value, 4)
        return view.copy() if copy else view

    def read_rgba8(self, rows: Optional[Tuple[int, int]] = None) -> np.ndarray:
        """The stored texels; ``rows`` = (first, count) reads that band of local rows only."""
        if rows is not None:
            out = np.empty((rows[1], self.size[0], 4), np.uint8)
            self.device.call("rt_read_rgba8_rows", rows[0], rows[1], out.ctypes.data_as(C.c_void_p), out.nbytes)
            return out
        out = np.empty((self._local_rows(), self.size[0], 4), np.uint8)
        self.device.call("rt_read_rgba8", out.ctypes.data_as(C.c_void_p), out.nbytes)
        return out

    def read_rgbf32(self, rows: Optional[Tuple[int, int]] = None) -> np.ndarray:
        if rows is not None:
            out = np.empty((rows[1], self.size[0], 3), np.float32)
            self.device.call("rt_read_rgbf32_rows", rows[0], rows[1], out.ctypes.data_as(C.c_void_p), out.nbytes)
            return out
        out = np.empty((self._local_rows(), self.size[0], 3), np.float32)
        self.device.call("rt_read_rgbf32", out.ctypes.data_as(C.c_void_p), out.nbytes)
        return out

    def read_segments(self) -> np.ndarray:
        rows = self._local_rows()
        out = np.empty((rows, self.size[0]), np.uint32)
        self.device.call("rt_read_segments", out.ctypes.data_as(C.c_void_p), out.nbytes)
        return out

    def read_hits(self, reflection_limit: int) -> np.ndarray:
        rows = self._local_rows()
        out = np.empty((rows, self.size[0], reflection_limit), np.int32)
        self.device.call("rt_read_hits", out.ctypes.data_as(C.c_void_p), out.nbytes)
        return out

    def write_history_rgba8(self, texels: np.ndarray):
        texels = np.ascontiguousarray(texels, np.uint8)
        self.device.call("rt_write_history_rgba8", texels.ctypes.data_as(C.c_void_p), texels.nbytes)


class RandomTexture:
    """random_texture.rs:1-102 — the 600×600 R32Float noise texture.

    ``start`` uploads a first texture and, with ``threaded=True``, re-generates and
    re-uploads it every ``interval`` seconds from a background thread exactly as the
    reference does (:56-61).  Unlike the reference the stream is seeded."""

    WIDTH, HEIGHT = 600, 600  # random_texture.rs:21-22

    def __init__(self, device: Device, seed: int = 1):
        self.device = device
        self.seed = seed
        self.data = None
        self._stop = threading.Event()
        self._thread = None

    @staticmethod
    def start(device: Device, seed: int = 1, threaded: bool = False, interval: float = 0.004) -> "RandomTexture":
        this = RandomTexture(device, seed)
        this.generate_data()
        this.write()
        if threaded:
            def loop():
                while not this._stop.wait(interval):
                    this.generate_data()
                    this.write()
            this._thread = threading.Thread(target=loop, daemon=True)
            this._thread.start()
        return this

    def generate_data(self):
        """random_texture.rs:67-75 (seeded SplitMix64 instead of fastrand)."""
        self.data = synth.random_texture(self.seed, self.WIDTH, self.HEIGHT)
        self.seed += 1

    def generate_on_device(self):
        """Same stream, generated by a kernel on the GPU (no host generation, no upload)."""
        rc = self.device._lib.rt_generate_random(self.device.handle, self.WIDTH, self.HEIGHT, self.seed)
        if rc != N.RT_OK:
            raise N.RtError(rc, "rt_generate_random")
        self.data = None
        self.seed += 1

    def read(self) -> np.ndarray:
        out = np.empty((self.HEIGHT, self.WIDTH), np.float32)
        self.device.call("rt_read_random", out.ctypes.data_as(C.c_void_p), out.nbytes)
        return out

    def write(self, data: Optional[np.ndarray] = None):
        """random_texture.rs:78-101."""
        if data is not None:
            self.data = np.ascontiguousarray(data, np.float32)
        h, w = self.data.shape
        # called from the generator thread: check the status without touching ctx->error
        rc = self.device._lib.rt_set_random(self.device.handle, w, h, self.data.ctypes.data_as(C.c_void_p))
        if rc != N.RT_OK:
            raise N.RtError(rc, "rt_set_random")

    def stop(self):
        self._stop.set()
        if self._thread is not None:
            self._thread.join()
            self._thread = None


class Connection:
    """connection.rs:42-411 — owns the scene/frame uniforms of the render path.

    ``max_objects=None`` keeps the reference's fixed 80-slot zero-padded object buffer
    (scene.rs:28, inputs.hlsl:77-80); an integer passes exactly that many wire records
    (``"live"`` = ``len(scene.objects)``, no padding)."""

    def __init__(self, scene: Scene, device: Device, skybox: np.ndarray,
                 random_texture: Optional[RandomTexture] = None, max_objects=None, jitter_seed: int = 42):
        self.device = device
        self.max_objects = max_objects
        self.last_scene = copy.deepcopy(scene)         # connection.rs:356
        self.last_size = (0, 0)                        # connection.rs:357
        self.frame_data = FrameData((0.0, 0.0))        # connection.rs:358
        self._jitter_rng = synth.SplitMix64(jitter_seed)
        self.load_hdri(skybox)
        self.random_texture = random_texture or RandomTexture.start(device)

    @staticmethod
    def load_image(file_name: str):
        """connection.rs:154-162 — decode an OpenEXR file to ((width, height), (H, W, 4) float32), alpha 1
        where the file has none (``into_rgba32f``).  Host code of the C-ABI library (rt_exr_read_rgba32f):
        single-part scan-line files, NONE / RLE / ZIPS / ZIP / PIZ; error texts as the reference words them."""
        from . import _native
        lib = _native.load()
        w, h = C.c_uint32(0), C.c_uint32(0)
        data = C.POINTER(C.c_float)()
        rc = lib.rt_exr_read_rgba32f(os.fsencode(file_name), C.byref(w), C.byref(h), C.byref(data))
        if rc != 0:
            raise _native.RtError(rc, "rt_exr_read_rgba32f", (lib.rt_exr_last_error() or b"").decode(errors="replace"))
        try:
            arr = np.ctypeslib.as_array(data, shape=(h.value, w.value, 4)).copy()
        finally:
            lib.rt_exr_free(data)
        return (w.value, h.value), arr

    def load_hdri(self, skybox):
        """connection.rs:176-215 — (H, W, 4) float32, row 0 = top, or the path of an .exr file (the
        reference reads ./assets/table_mountain_1_8k.exr)."""
        if isinstance(skybox, (str, bytes, os.PathLike)):  # decode + upload inside the library (rt_set_skybox_exr)
            rc = self.device._lib.rt_set_skybox_exr(self.device.handle, os.fsencode(os.fspath(skybox)))
            if rc != N.RT_OK:
                text = self.device._lib.rt_exr_last_error() or self.device._lib.rt_last_error(self.device.handle) or b""
                raise N.RtError(rc, "rt_set_skybox_exr", text.decode(errors="replace"))
            return
        skybox = np.ascontiguousarray(skybox, np.float32)
        if skybox.ndim != 3 or skybox.shape[2] != 4:
            raise ValueError("skybox must be (H, W, 4) float32")
        self.device.call("rt_set_skybox", skybox.shape[1], skybox.shape[0], skybox.ctypes.data_as(C.c_void_p))

    def _object_blob(self, scene: Scene, width: int, height: int):
        if self.max_objects == "live":
            return scene.live_object_bytes(), len(scene.objects)
        slots = Scene.MAX_OBJECTS if self.max_objects is None else int(self.max_objects)
        return scene.as_bytes(width, height, slots)[0], slots

    def update_buffers(self, size: Tuple[int, int], scene: Scene, jitter: Optional[Tuple[float, float]] = None):
        """connection.rs:370-410.  ``jitter`` None draws r·0.99 − 0.495 per axis from the
        connection's seeded stream (the reference uses ``rand::random``, :392-399)."""
        size = tuple(size)
        if (scene != self.last_scene) | (size != self.last_size):  # connection.rs:373
            objects, count = self._object_blob(scene, size[0], size[1])
            lights = scene.as_bytes(size[0], size[1])[1]
            config = scene.config_bytes(size[0], size[1])
            self.device.call("rt_set_scene", objects, count, lights, Scene.MAX_LIGHTS, config)
            self.last_scene = copy.deepcopy(scene)
            self.last_size = size
            self.frame_data.progressive_count = 0
        else:
            self.frame_data.progressive_count += 1
        if jitter is None:
            s = FrameData.JITTER_STRENGTH
            half = -(np.float32(s) / np.float32(2.0))  # mul_add: one rounding (connection.rs:392-399)
            jitter = (float(fma_f32(self._jitter_rng.f32(), np.float32(s), half)),
                      float(fma_f32(self._jitter_rng.f32(), np.float32(s), half)))
        self.frame_data.jitter = (float(jitter[0]), float(jitter[1]))

    def draw(self):
        """The render pass + history copy + submit of app.rs:219-250."""
        self.device.call("rt_render_frame", self.frame_data.as_bytes())

    def draw_frames(self, jitters, first_progressive_count: Optional[int] = None):
        """len(jitters) consecutive frames of the progressive accumulation in ONE launch
        (rt_render_frames): frame i has jitter jitters[i] and progressive_count first + i, where
        first defaults to this connection's current count.  Bit-identical to calling draw() once
        per frame; leaves frame_data at the last frame of the batch."""
        first = self.frame_data.progressive_count if first_progressive_count is None else int(first_progressive_count)
        blob = bytearray()
        for i, j in enumerate(jitters):
            self.frame_data.jitter = (float(j[0]), float(j[1]))
            self.frame_data.progressive_count = first + i
            blob += self.frame_data.as_bytes()
        if not blob:
            return
        self.device.call("rt_render_frames", len(blob) // FrameData.BUFFER_SIZE, bytes(blob))


def render_scene(device: Device, connection: Connection, render_target: RenderTarget, scene: Scene,
                 jitter: Optional[Tuple[float, float]] = None):
    """app.rs:208-254: update_buffers → draw; asynchronous (``device.sync()`` to wait)."""
    connection.update_buffers(render_target.size, scene, jitter)
    connection.draw()
