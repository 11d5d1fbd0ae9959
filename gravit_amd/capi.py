"""ctypes binding of libgvt_hip.so (include/gvt_hip.h) -- the only way the Python host side reaches
the device.  There is NO CPU fallback: a missing library or a failing call raises.
"""
import ctypes as C
import os

import numpy as np

from .layouts import HIT_DTYPE, LIGHT_DTYPE, MATERIAL_DTYPE, RAY_DTYPE

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GVT_HIP_LIB") or os.path.join(HERE, "libgvt_hip.so")  # GVT_HIP_LIB: an alternative build (A/B measurements)

# every symbol include/gvt_hip.h declares
ABI_VERSION = 6  # GVT_HIP_ABI_VERSION of include/gvt_hip.h this file mirrors (tests/test_host_cpu.py compares the two)

SYMBOLS = [
    "gvt_hip_init", "gvt_hip_set_stream", "gvt_hip_synchronize", "gvt_hip_last_error",
    "gvt_hip_mesh_create", "gvt_hip_mesh_destroy", "gvt_hip_trace_ex", "gvt_hip_mesh_get_info", "gvt_hip_mesh_get_normals",
    "gvt_hip_trace", "gvt_hip_intersect", "gvt_hip_occluded",
    "gvt_hip_queue_create", "gvt_hip_queue_destroy", "gvt_hip_queue_reserve", "gvt_hip_queue_clear", "gvt_hip_queue_size",
    "gvt_hip_queue_append", "gvt_hip_queue_append_flags", "gvt_hip_abi_version", "gvt_hip_queue_export", "gvt_hip_trace_queue", "gvt_hip_trace_queue_sink",
    "gvt_hip_camera_generate",
    "gvt_hip_camera_generate_tiled",
    "gvt_hip_camera_filter",
    "gvt_hip_top_create", "gvt_hip_top_destroy", "gvt_hip_top_order", "gvt_hip_shuffle", "gvt_hip_queue_sizes",
    "gvt_hip_fb_create", "gvt_hip_fb_destroy", "gvt_hip_fb_clear", "gvt_hip_fb_download", "gvt_hip_fb_device_ptr",
    "gvt_hip_fb_write_ppm_bytes",
    "gvt_hip_profile", "gvt_hip_stats_read", "gvt_hip_stats_reset", "gvt_hip_set_option", "gvt_hip_is_experiments_build", "gvt_hip_counters_peek", "gvt_hip_visit_stats", "gvt_hip_wide_visit_stats", "gvt_hip_mesh_download_nodes", "gvt_hip_mesh_download_wide", "gvt_hip_mesh_download_clusters", "gvt_hip_mesh_upload_nodes", "gvt_hip_marked_visit_stats", "gvt_hip_image_frame",
    "gvt_hip_math_probe", "gvt_hip_ctx_create", "gvt_hip_ctx_make_current", "gvt_hip_ctx_destroy",
    "gvt_hip_comm_unique_id", "gvt_hip_comm_create", "gvt_hip_hub_create", "gvt_hip_hub_abort", "gvt_hip_hub_destroy", "gvt_hip_comm_create_local",
    "gvt_hip_comm_destroy", "gvt_hip_comm_rank", "gvt_hip_comm_world", "gvt_hip_comm_count", "gvt_hip_comm_reserved_cus", "gvt_hip_comm_set_deadline_ms", "gvt_hip_comm_selftest",
    "gvt_hip_tracer_create", "gvt_hip_tracer_destroy", "gvt_hip_tracer_set_camera", "gvt_hip_tracer_set_domains", "gvt_hip_tracer_frame",
]


class GvtHipError(RuntimeError):
    pass


class MeshInfo(C.Structure):
    _fields_ = [("n_tris", C.c_uint64), ("n_verts", C.c_uint64), ("n_nodes", C.c_uint64), ("n_leaves", C.c_uint64),
                ("bbox_lo", C.c_float * 3), ("bbox_hi", C.c_float * 3), ("build_ms", C.c_float), ("max_leaf", C.c_uint32),
                ("packet", C.c_uint32), ("bytes_nodes", C.c_uint64), ("bytes_tris", C.c_uint64), ("sah_inner", C.c_float), ("pad", C.c_float)]


class CameraPod(C.Structure):
    _fields_ = [("eye", C.c_float * 3), ("focus", C.c_float * 3), ("up", C.c_float * 3), ("fov", C.c_float), ("width", C.c_int32),
                ("height", C.c_int32), ("samples", C.c_int32), ("depth", C.c_int32), ("jitter_window_size", C.c_float)]


class FrameStats(C.Structure):
    _fields_ = [("rounds", C.c_uint64), ("chains", C.c_uint64), ("host_syncs", C.c_uint64), ("rays_sent", C.c_uint64),
                ("rays_closest", C.c_uint64), ("rays_any", C.c_uint64), ("packets_bailed", C.c_uint64), ("bytes_sent", C.c_uint64),
                ("ms_chain", C.c_double), ("ms_announce", C.c_double), ("ms_payload", C.c_double), ("ms_composite", C.c_double), ("ms_host_wait", C.c_double),
                ("exchanges", C.c_uint64), ("rays_inline", C.c_uint64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


FRAME_BSP, FRAME_NO_COMPOSITE, FRAME_FULL_REDUCE, FRAME_IMAGE = 1, 2, 4, 8


class Stats(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_any", C.c_uint64), ("rays_shaded", C.c_uint64),
                ("rays_forwarded", C.c_uint64), ("trace_calls", C.c_uint64),
                ("ms_closest", C.c_double), ("ms_any", C.c_double), ("ms_shade", C.c_double), ("ms_convert", C.c_double),
                ("ms_shuffle", C.c_double), ("ms_camera", C.c_double), ("ms_build", C.c_double),
                ("launches_closest", C.c_uint64), ("launches_any", C.c_uint64), ("ms_sort", C.c_double), ("ms_long", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_lib = None


def load():
    """Load the library (no device call).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GvtHipError("%s is missing: build it with `python -m gravit_amd._build` (hipcc, gfx950). "
                              "There is no CPU fallback for the adapter." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for s in SYMBOLS:
            getattr(lib, s)  # AttributeError if the ABI is incomplete
        lib.gvt_hip_last_error.restype = C.c_char_p
        if lib.gvt_hip_abi_version() != ABI_VERSION:  # the out-structs below (MeshInfo, Stats, FrameStats) mirror ONE revision of include/gvt_hip.h
            raise GvtHipError("%s is ABI revision %d, this binding was written against %d: rebuild the library (python -m gravit_amd._build)" % (LIB_PATH, lib.gvt_hip_abi_version(), ABI_VERSION))
        for f in ("gvt_hip_mesh_create", "gvt_hip_queue_create", "gvt_hip_top_create", "gvt_hip_fb_create", "gvt_hip_fb_device_ptr", "gvt_hip_ctx_create",
                  "gvt_hip_comm_create", "gvt_hip_hub_create", "gvt_hip_comm_create_local", "gvt_hip_tracer_create"):
            getattr(lib, f).restype = C.c_void_p
        for f in ("gvt_hip_mesh_destroy", "gvt_hip_queue_destroy", "gvt_hip_top_destroy", "gvt_hip_fb_destroy", "gvt_hip_ctx_destroy", "gvt_hip_hub_abort",
                  "gvt_hip_hub_destroy", "gvt_hip_comm_destroy", "gvt_hip_tracer_destroy"):
            getattr(lib, f).restype = None
            getattr(lib, f).argtypes = [C.c_void_p]
        _lib = lib
    return _lib


def last_error():
    return load().gvt_hip_last_error().decode(errors="replace")


def check(rc, what):
    if rc != 0:
        raise GvtHipError("%s failed (%d): %s" % (what, rc, last_error()))


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a if shape is None else a.reshape(shape)


def counters_peek():
    """Diagnostic: the 32 counter words of the calling thread's context after a stream synchronisation ([3] rays the last closest-hit
    launch parked for k_long_closest, [8] overflow flags)."""
    out = (C.c_uint32 * 32)()
    check(load().gvt_hip_counters_peek(out), "gvt_hip_counters_peek")
    return list(out)


def init(device=0):
    check(load().gvt_hip_init(C.c_int(device)), "gvt_hip_init")


def set_stream(stream_handle):
    check(load().gvt_hip_set_stream(C.c_void_p(stream_handle)), "gvt_hip_set_stream")


def synchronize():
    check(load().gvt_hip_synchronize(), "gvt_hip_synchronize")


def profile(enable):
    check(load().gvt_hip_profile(C.c_int(int(enable))), "gvt_hip_profile")


def stats(reset=False):
    s = Stats()
    check(load().gvt_hip_stats_read(C.byref(s)), "gvt_hip_stats_read")
    if reset:
        check(load().gvt_hip_stats_reset(), "gvt_hip_stats_reset")
    return s.as_dict()


def set_option(name, value):
    check(load().gvt_hip_set_option(name.encode(), C.c_int(int(value))), "gvt_hip_set_option")


def math_probe(kind, x):
    """include/gvt_math.h evaluated on the device (diagnostic): kind 0 gvt_sinf, 1 gvt_cosf, 2 (float)gvt_acos(sqrt(1 - x))."""
    x = f32(x, -1)
    out = np.zeros_like(x)
    check(load().gvt_hip_math_probe(C.c_int(kind), ptr(x), C.c_size_t(len(x)), ptr(out)), "gvt_hip_math_probe")
    return out


def stats_reset():
    check(load().gvt_hip_stats_reset(), "gvt_hip_stats_reset")


__all__ = ["load", "init", "check", "ptr", "f32", "GvtHipError", "MeshInfo", "Stats", "SYMBOLS", "RAY_DTYPE", "HIT_DTYPE",
           "LIGHT_DTYPE", "MATERIAL_DTYPE"]
