"""ctypes binding of ``librt355.so`` (C-ABI in ``include/rt355.h``).

There is no fallback: if the shared library is missing or no gfx950 device can be
opened, every entry point raises.  The library is built in-tree by
``__graft_entry__.build()`` / ``make -C rsraytracingv3_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("RT355_LIB", os.path.join(_HERE, "librt355.so"))  # override: A/B of two builds

RT_OK = 0
RT_ACCUM_SRGB8, RT_ACCUM_FLOAT32 = 0, 1
RT_OUT_RGBA8, RT_OUT_RGBF32, RT_OUT_SEGMENTS, RT_OUT_HITS = 1, 2, 4, 8
RT_OPT_VARIANT, RT_OPT_REFILL_MIN, RT_OPT_BLOCKS_PER_CU, RT_OPT_TILE_CHUNK, RT_OPT_FLAGS, RT_OPT_CULL = 0, 1, 2, 3, 4, 5
RT_OPT_INFLIGHT, RT_OPT_GRID_MIN, RT_OPT_BATCH_INFLIGHT = 6, 7, 8
RT_OPT_DIRMASK, RT_OPT_DIRMASK_MIN, RT_OPT_DIRMASK_CELLS, RT_OPT_DIRMASK_K, RT_OPT_TAIL_SPILL = 9, 10, 11, 12, 13
RT355_PRESENT_SLOTS = 4
RT_TAIL_SPILL_DEFAULT = -1  # the default of the scene's kind (kDefaultTailSpill 32, kDefaultTailSpillLists 24: csrc/rt_device.h)
RT_OPT_TAIL_SPILL_SINGLE = 14  # default 32 (single frames of 24 000 tiles or more and 7 bounces or more; 5 with candidate lists)
RT_TAIL_SPILL_SINGLE_DEFAULT = 32
RT_OPT_TIMING = 20
RT_OPT_DIRLIST, RT_OPT_DIRLIST_MIN, RT_OPT_DIRLIST_CELLS, RT_OPT_DIRLIST_K, RT_OPT_DIRLIST_REACH = 15, 16, 17, 18, 19

STATUS_NAMES = {0: "RT_OK", -1: "RT_ERR_INVALID_ARGUMENT", -2: "RT_ERR_NO_DEVICE", -3: "RT_ERR_NOT_READY",
                -4: "RT_ERR_DEVICE", -5: "RT_ERR_OUT_OF_MEMORY"}


class RtError(RuntimeError):
    def __init__(self, code: int, where: str, text: str = ""):
        self.code = code
        super().__init__("%s: %s%s" % (where, STATUS_NAMES.get(code, str(code)), (" — " + text) if text else ""))


class RtStats(C.Structure):
    _fields_ = [("frames", C.c_uint64), ("last_segments", C.c_uint64), ("last_rough_hits", C.c_uint64),
                ("last_pixels", C.c_uint64), ("last_kernel_ms", C.c_float), ("width", C.c_uint32),
                ("height", C.c_uint32), ("local_rows", C.c_uint32), ("object_count", C.c_uint32),
                ("sphere_count", C.c_uint32), ("plane_count", C.c_uint32), ("last_batch", C.c_uint32),
                ("last_wave_iterations", C.c_uint64), ("last_cull_mismatches", C.c_uint64),
                ("last_resolve_ms", C.c_float), ("last_variant", C.c_uint32), ("last_stage", C.c_uint32),
                ("grid_cells", C.c_uint32), ("grid_items", C.c_uint32),
                ("dirmask_cells", C.c_uint32), ("dirmask_bins", C.c_uint32), ("tail_spill_unavailable", C.c_uint32),
                ("path_queue_bytes", C.c_uint64), ("dirmask_lists", C.c_uint32), ("spheres_scanned", C.c_uint32),
                ("last_launches_in_flight", C.c_uint32), ("reserved0", C.c_uint32)]


class RtGridInfo(C.Structure):
    _fields_ = [("enabled", C.c_uint32), ("nx", C.c_uint32), ("ny", C.c_uint32), ("nz", C.c_uint32), ("n_cells", C.c_uint32),
                ("n_items", C.c_uint32), ("gmin", C.c_float * 3), ("gmax", C.c_float * 3), ("cell", C.c_float * 3),
                ("inv_cell", C.c_float * 3), ("near_min", C.c_float * 3), ("near_max", C.c_float * 3), ("t_margin", C.c_float)]


class RtDirMaskInfo(C.Structure):
    _fields_ = [("enabled", C.c_uint32), ("nx", C.c_uint32), ("ny", C.c_uint32), ("nz", C.c_uint32), ("k", C.c_uint32),
                ("bins", C.c_uint32), ("words", C.c_uint32), ("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("cell", C.c_float * 3),
                ("inv_cell", C.c_float * 3), ("near_min", C.c_float * 3), ("near_max", C.c_float * 3),
                ("n_outliers", C.c_uint32), ("outlier_slot", C.c_uint32 * 4)]


class RtDirListInfo(C.Structure):
    _fields_ = [("enabled", C.c_uint32), ("nx", C.c_uint32), ("ny", C.c_uint32), ("nz", C.c_uint32), ("k", C.c_uint32),
                ("bins", C.c_uint32), ("bmin", C.c_float * 3), ("bmax", C.c_float * 3), ("cell", C.c_float * 3),
                ("inv_cell", C.c_float * 3), ("reach_unit", C.c_float), ("s_cap", C.c_float),
                ("n_outliers", C.c_uint32), ("outlier_slot", C.c_uint32 * 4)]


class RtDeviceBuffers(C.Structure):
    _fields_ = [("rgba8", C.c_void_p), ("rgbf32", C.c_void_p), ("segments", C.c_void_p), ("hits", C.c_void_p),
                ("stream", C.c_void_p), ("width", C.c_uint32), ("local_rows", C.c_uint32)]


# name -> (restype, argtypes); every symbol include/rt355.h declares
_vp, _u32, _i, _f, _sz = C.c_void_p, C.c_uint32, C.c_int, C.c_float, C.c_size_t
SIGNATURES = {
    "rt_abi_version": (_u32, []),
    "rt_has_ab_shapes": (_i, []),
    "rt_create": (_i, [_i, C.POINTER(_vp)]),
    "rt_destroy": (_i, [_vp]),
    "rt_set_skybox": (_i, [_vp, _u32, _u32, _vp]),
    "rt_exr_read_rgba32f": (_i, [C.c_char_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_float))]),
    "rt_exr_free": (None, [C.POINTER(C.c_float)]),
    "rt_exr_last_error": (C.c_char_p, []),
    "rt_set_skybox_exr": (_i, [_vp, C.c_char_p]),
    "rt_set_random": (_i, [_vp, _u32, _u32, _vp]),
    "rt_generate_random": (_i, [_vp, _u32, _u32, C.c_uint64]),
    "rt_read_random": (_i, [_vp, _vp, _sz]),
    "rt_set_scene": (_i, [_vp, _vp, _u32, _vp, _u32, _vp]),
    "rt_resize": (_i, [_vp, _u32, _u32]),
    "rt_set_partition": (_i, [_vp, _u32, _u32, _u32]),
    "rt_grid_build_host": (_i, [_vp, _u32, _u32, C.POINTER(RtGridInfo), _vp, _u32, _vp, _u32]),
    "rt_dirmask_build_host": (_i, [_vp, _u32, _u32, _u32, C.POINTER(RtDirMaskInfo), _vp, _sz]),
    "rt_dirmask_lookup_host": (_i, [C.POINTER(RtDirMaskInfo), _vp, _vp, _vp, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]),
    "rt_dirlist_plan_host": (_i, [_vp, _u32, _u32, _u32, _u32, C.POINTER(RtDirListInfo), _vp]),
    "rt_dirlist_entry_host": (_i, [C.POINTER(RtDirListInfo), _vp, _u32, _u32, _u32, _vp]),
    "rt_dirlist_walk_host": (_i, [C.POINTER(RtDirListInfo), _vp, _vp, _u32, _vp, _vp, _f, C.POINTER(C.c_uint32), C.POINTER(C.c_int32),
                                  C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_int)]),
    "rt_comm_available": (_i, [C.c_char_p, _sz]),
    "rt_comm_get_unique_id": (_i, [_vp]),
    "rt_comm_init": (_i, [_vp, _vp, _u32, _u32, _u32]),
    "rt_comm_destroy": (_i, [_vp]),
    "rt_comm_idle": (_i, [_vp, _vp]),
    "rt_comm_abort": (_i, [_vp]),
    "rt_read_frame_rgba8": (_i, [_vp, _vp, _sz]),
    "rt_multi_create": (_i, [C.POINTER(C.c_int), _u32, _u32, C.POINTER(_vp)]),
    "rt_multi_destroy": (_i, [_vp]),
    "rt_multi_device_count": (_u32, [_vp]),
    "rt_multi_peer_status": (_i, [_vp, C.POINTER(_u32), C.POINTER(_u32)]),
    "rt_multi_context": (_vp, [_vp, _u32]),
    "rt_multi_last_error": (C.c_char_p, [_vp]),
    "rt_multi_set_skybox": (_i, [_vp, _u32, _u32, _vp]),
    "rt_multi_set_skybox_exr": (_i, [_vp, C.c_char_p]),
    "rt_multi_set_random": (_i, [_vp, _u32, _u32, _vp]),
    "rt_multi_generate_random": (_i, [_vp, _u32, _u32, C.c_uint64]),
    "rt_multi_set_scene": (_i, [_vp, _vp, _u32, _vp, _u32, _vp]),
    "rt_multi_resize": (_i, [_vp, _u32, _u32]),
    "rt_multi_set_accum_mode": (_i, [_vp, _i]),
    "rt_multi_set_option": (_i, [_vp, _i, _i]),
    "rt_multi_render": (_i, [_vp, _f, _f, _u32]),
    "rt_multi_render_frame": (_i, [_vp, _vp]),
    "rt_multi_render_frames": (_i, [_vp, C.c_uint32, _vp]),
    "rt_multi_sync": (_i, [_vp]),
    "rt_multi_read_frame_rgba8": (_i, [_vp, _vp, _sz]),
    "rt_multi_get_stats": (_i, [_vp, C.POINTER(RtStats)]),
    "rt_assemble_strips": (_i, [_vp, _u32, _u32, _u32, _u32, _vp, _vp, _vp]),
    "rt_set_accum_mode": (_i, [_vp, _i]),
    "rt_set_outputs": (_i, [_vp, _u32]),
    "rt_set_variant": (_i, [_vp, _i]),
    "rt_set_option": (_i, [_vp, _i, _i]),
    "rt_selftest_sqrt": (_i, [_vp, C.POINTER(C.c_uint64)]),
    "rt_render": (_i, [_vp, _f, _f, _u32]),
    "rt_render_frame": (_i, [_vp, _vp]),
    "rt_render_frames": (_i, [_vp, C.c_uint32, _vp]),
    "rt_keep_batch_frames": (_i, [_vp, _u32]),
    "rt_read_batch_frame": (_i, [_vp, _u32, _vp, _sz]),
    "rt_sync": (_i, [_vp]),
    "rt_read_rgba8": (_i, [_vp, _vp, _sz]),
    "rt_read_rgbf32": (_i, [_vp, _vp, _sz]),
    "rt_read_rgba8_rows": (_i, [_vp, _u32, _u32, _vp, _sz]),
    "rt_read_rgbf32_rows": (_i, [_vp, _u32, _u32, _vp, _sz]),
    "rt_read_segments": (_i, [_vp, _vp, _sz]),
    "rt_read_hits": (_i, [_vp, _vp, _sz]),
    "rt_present_async": (_i, [_vp, _u32]),
    "rt_present_wait": (_i, [_vp, _u32, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_u32), C.POINTER(_u32)]),
    "rt_multi_present_async": (_i, [_vp, _u32]),
    "rt_multi_present_wait": (_i, [_vp, _u32, C.POINTER(_vp), C.POINTER(_sz), C.POINTER(_u32), C.POINTER(_u32)]),
    "rt_write_history_rgba8": (_i, [_vp, _vp, _sz]),
    "rt_get_device_buffers": (_i, [_vp, C.POINTER(RtDeviceBuffers)]),
    "rt_get_stats": (_i, [_vp, C.POINTER(RtStats)]),
    "rt_get_counter_words": (_i, [_vp, _vp]),
    "rt_get_frame_history": (_i, [_vp, _u32, _vp, _vp, _vp]),
    "rt_get_resolve_history": (_i, [_vp, _u32, _vp]),
    "rt_get_tail_history": (_i, [_vp, _u32, _vp]),
    "rt_get_launch_span": (_i, [_vp, _u32, _vp]),
    "rt_set_stream": (_i, [_vp, _vp]),
    "rt_bind_outputs": (_i, [_vp, _vp, _vp, _vp]),
    "rt_last_error": (C.c_char_p, [_vp]),
    "rt_selftest_math_unary": (_i, [_vp, _i, _vp, _vp, _sz]),
    "rt_selftest_math_binary": (_i, [_vp, _i, _vp, _vp, _vp, _sz]),
}

_lib = None


def kernel_source_hash() -> str:
    """sha256 over the sources that decide what the render kernels do and how they are launched (csrc/rt_kernels.hip,
    rt_device.h — which also holds the defaults of the launch-shaping options —, rt_math.h, rt_dirmask.h, rt_grid.h,
    Makefile), in name order.  Profiles record it
    (tools/pmc_summary.py) and bench.py only quotes a profile's counters when the tree it runs from still has the sources
    the profile was measured on."""
    import hashlib
    h = hashlib.sha256()
    names = ("Makefile", "rt_device.h", "rt_dirmask.h", "rt_grid.h", "rt_kernels.hip", "rt_math.h")
    files = [os.path.join(_HERE, "csrc", n) for n in names]
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _preload_torch_hip_runtime():
    """One HIP runtime per process.  PyTorch wheels carry their own libamdhip64/libhsa-runtime64; a process
    that first loads librt355.so (which would pull in /opt/rocm's copy) and then imports torch ends up with
    two runtimes, and the second one finds no GPU ("No HIP GPUs are available").  When torch is installed,
    its copy is loaded first — without importing torch — so both sides share it, whatever the import order."""
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        return
    if spec is None or not spec.origin:
        return
    candidate = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(candidate):
        C.CDLL(candidate, mode=C.RTLD_GLOBAL)


def load():
    """Load ``librt355.so``; raises (never falls back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "rt355: %s not found — the HIP extension is not built.  Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C rsraytracingv3_amd/csrc`. "
                "There is no CPU fallback." % LIB_PATH)
        _preload_torch_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        host_only = os.environ.get("RT355_HOST_ONLY") == "1"  # the sanitizer build of the host code (make host-asan): no device entry points
        for name, (res, args) in SIGNATURES.items():
            if host_only and not hasattr(lib, name):
                continue
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib
