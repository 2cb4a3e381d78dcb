"""ctypes front-end of the CPU oracle (``oracle/rt_oracle.c``).

TEST INFRASTRUCTURE ONLY — imported by ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never by the ``rsraytracingv3_amd`` package.
Parity status of the oracle itself: pinned on geometry and emission by the reference's screenshot
(``tests/test_screenshot_pin.py``), texel values otherwise *unpinned* (the reference has no tests or golden
vectors and cannot be built here — see the header of ``rt_oracle.c``).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "librt_oracle.so")
_lib = None


def _host_has_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            return " fma " in f.read().replace("\n", " ")
    except OSError:
        return False


def build(force: bool = False) -> str:
    """Compile ``librt_oracle.so`` with the committed Makefile (gcc, plain C)."""
    srcs = [os.path.join(_HERE, n) for n in ("rt_oracle.c", "rt_oracle_math.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs)
    if force or stale:
        args = ["make", "-C", _HERE, "-B", "librt_oracle.so"]
        if not _host_has_fma():
            args.append("FMAFLAG=")
        subprocess.run(args, check=True, capture_output=True)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u8p, f32p, u32p, i32p = (C.POINTER(t) for t in (C.c_uint8, C.c_float, C.c_uint32, C.c_int32))
        L.rto_render.restype = C.c_int
        L.rto_render.argtypes = [u8p, C.c_uint32, u8p, u8p, f32p, C.c_uint32, C.c_uint32, f32p, C.c_uint32,
                                 C.c_uint32, u8p, f32p, C.c_int, f32p, u8p, u32p, i32p, C.c_uint32, C.c_uint32,
                                 C.c_int]
        L.rto_set_ray_dump.restype = None
        L.rto_set_ray_dump.argtypes = [f32p]
        L.rto_probe_intersect.restype = C.c_int
        L.rto_probe_intersect.argtypes = [u8p, C.c_uint32, f32p, f32p, f32p]
        L.rto_probe_create_ray.restype = None
        L.rto_probe_create_ray.argtypes = [u8p, C.c_float, C.c_float, f32p]
        L.rto_probe_random.restype = None
        L.rto_probe_random.argtypes = [f32p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_uint32, f32p, f32p]
        L.rto_probe_hdri_texel.restype = None
        L.rto_probe_hdri_texel.argtypes = [f32p, C.c_uint32, C.c_uint32, u32p]
        L.rto_probe_hemisphere.restype = None
        L.rto_probe_hemisphere.argtypes = [f32p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, f32p, C.c_float, f32p]
        L.rto_srgb8_encode.restype = C.c_uint8
        L.rto_srgb8_encode.argtypes = [C.c_float]
        L.rto_srgb8_decode.restype = C.c_float
        L.rto_srgb8_decode.argtypes = [C.c_uint32]
        L.rto_math_unary.restype = None
        L.rto_math_unary.argtypes = [C.c_int, f32p, f32p, C.c_size_t]
        L.rto_math_binary.restype = None
        L.rto_math_binary.argtypes = [C.c_int, f32p, f32p, f32p, C.c_size_t]
        L.rto_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _p(a: Optional[np.ndarray], t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def _u8(b) -> np.ndarray:
    return np.frombuffer(bytes(b), dtype=np.uint8).copy()


class Frame:
    """Outputs of one oracle frame."""

    def __init__(self, rgbf32, rgba8, segments, hits):
        self.rgbf32, self.rgba8, self.segments, self.hits = rgbf32, rgba8, segments, hits


def render(objects: bytes, object_count: int, config112: bytes, jitter=(0.0, 0.0), progressive_count: int = 0,
           hdri: np.ndarray = None, random: np.ndarray = None, history_rgba8: Optional[np.ndarray] = None,
           history_f32: Optional[np.ndarray] = None, accum_mode: int = 0, rows=None, threads: int = 0,
           want_hits: bool = True) -> Frame:
    """Run ``rto_render``.  ``hdri`` is (Ht, Wt, 4) float32, ``random`` (Hr, Wr) float32."""
    import struct
    cfg = _u8(config112)
    w, h = struct.unpack_from("<II", bytes(config112), 100)
    rlimit = struct.unpack_from("<I", bytes(config112), 96)[0]
    fd = _u8(struct.pack("<ffII", float(jitter[0]), float(jitter[1]), int(progressive_count), 0))
    obj = _u8(objects) if object_count else np.zeros(1, np.uint8)
    assert len(objects) >= 112 * object_count
    hdri = np.ascontiguousarray(hdri, dtype=np.float32)
    random = np.ascontiguousarray(random, dtype=np.float32)
    out_f = np.zeros((h, w, 3), np.float32)
    out_8 = np.zeros((h, w, 4), np.uint8)
    out_s = np.zeros((h, w), np.uint32)
    out_h = np.full((h, w, max(rlimit, 1)), -2, np.int32) if want_hits else None
    if history_rgba8 is not None:
        history_rgba8 = np.ascontiguousarray(history_rgba8, dtype=np.uint8)
    if history_f32 is not None:
        history_f32 = np.ascontiguousarray(history_f32, dtype=np.float32)
    y0, y1 = (0, h) if rows is None else rows
    rc = lib().rto_render(_p(obj, C.c_uint8), object_count, _p(cfg, C.c_uint8), _p(fd, C.c_uint8),
                          _p(hdri, C.c_float), hdri.shape[1], hdri.shape[0],
                          _p(random, C.c_float), random.shape[1], random.shape[0],
                          _p(history_rgba8, C.c_uint8), _p(history_f32, C.c_float), accum_mode,
                          _p(out_f, C.c_float), _p(out_8, C.c_uint8), _p(out_s, C.c_uint32),
                          _p(out_h, C.c_int32), y0, y1, threads)
    if rc != 0:
        raise RuntimeError("rto_render failed: %d" % rc)
    return Frame(out_f, out_8, out_s, out_h)


def render_with_rays(objects: bytes, object_count: int, config112: bytes, *args, **kwargs):
    """``render`` + the ray (origin, direction) of every executed iteration: (Frame, rays[h, w, reflection_limit, 6]),
    NaN where an iteration did not run.  Diagnostic for tests of the product's culling structures."""
    import struct
    cfg = bytes(config112)
    w, h = struct.unpack_from("<II", cfg, 100)
    rlimit = struct.unpack_from("<I", cfg, 96)[0]
    rays = np.full((h, w, max(rlimit, 1), 6), np.nan, np.float32)
    lib().rto_set_ray_dump(_p(rays, C.c_float))
    try:
        frame = render(objects, object_count, config112, *args, **kwargs)
    finally:
        lib().rto_set_ray_dump(None)
    return frame, rays


def probe_intersect(objects: bytes, object_count: int, origin, direction):
    obj = _u8(objects)
    o = np.asarray(origin, np.float32)
    d = np.asarray(direction, np.float32)
    out = np.zeros(7, np.float32)
    idx = lib().rto_probe_intersect(_p(obj, C.c_uint8), object_count, _p(o, C.c_float), _p(d, C.c_float),
                                    _p(out, C.c_float))
    return idx, float(out[0]), out[1:4].copy(), out[4:7].copy()


def probe_create_ray(config112: bytes, u: float, v: float) -> np.ndarray:
    cfg = _u8(config112)
    out = np.zeros(3, np.float32)
    lib().rto_probe_create_ray(_p(cfg, C.c_uint8), u, v, _p(out, C.c_float))
    return out


def probe_random(random: np.ndarray, px: float, py: float, n: int):
    random = np.ascontiguousarray(random, np.float32)
    vals, seeds = np.zeros(n, np.float32), np.zeros(n, np.float32)
    lib().rto_probe_random(_p(random, C.c_float), random.shape[1], random.shape[0], px, py, n,
                           _p(vals, C.c_float), _p(seeds, C.c_float))
    return vals, seeds


def probe_hdri_texel(direction, w: int, h: int):
    d = np.asarray(direction, np.float32)
    out = np.zeros(2, np.uint32)
    lib().rto_probe_hdri_texel(_p(d, C.c_float), w, h, _p(out, C.c_uint32))
    return int(out[0]), int(out[1])


def probe_hemisphere(random: np.ndarray, px: float, py: float, normal, roughness: float) -> np.ndarray:
    random = np.ascontiguousarray(random, np.float32)
    n = np.asarray(normal, np.float32)
    out = np.zeros(3, np.float32)
    lib().rto_probe_hemisphere(_p(random, C.c_float), random.shape[1], random.shape[0], px, py,
                               _p(n, C.c_float), roughness, _p(out, C.c_float))
    return out


_UNARY = {"sin": 0, "cos": 1, "tan": 2, "asin": 3, "log2": 4, "exp2": 5}
_BINARY = {"atan2": 0, "pow": 1}


def math_unary(name: str, x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    lib().rto_math_unary(_UNARY[name], _p(x, C.c_float), _p(y, C.c_float), x.size)
    return y


def math_binary(name: str, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    y = np.empty_like(a)
    lib().rto_math_binary(_BINARY[name], _p(a, C.c_float), _p(b, C.c_float), _p(y, C.c_float), a.size)
    return y


def srgb8_encode(c: float) -> int:
    return int(lib().rto_srgb8_encode(c))


def srgb8_decode(b: int) -> float:
    return float(lib().rto_srgb8_decode(b))


def max_threads() -> int:
    return int(lib().rto_max_threads())
