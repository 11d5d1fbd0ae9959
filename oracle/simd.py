"""ctypes binding of oracle/libsimd_baseline.so -- the SIMD CPU baseline that bench.py's `cpu_baseline` leg times beside the GPU number (one ray
against four quantised child boxes per step in SSE, over the GPU-built 4-wide tree downloaded once: oracle/simd_baseline.c).  Measurement
infrastructure: only bench.py's cpu_baseline leg and tests/ may import it; it is validated against the oracle (orc.Mesh.intersect / occluded)."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(os.path.join(_HERE, "libsimd_baseline.so"))
        _lib.simd_intersect.restype = C.c_int
        _lib.simd_occluded.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Tree:
    """the traversal layout of one mesh as gvt_hip_mesh_download_wide returns it (HipMeshAdapter.download_wide())"""

    def __init__(self, nodes4, slots):
        self.nodes4 = np.ascontiguousarray(nodes4, np.uint32)
        self.slots = np.ascontiguousarray(slots, np.float32)

    def intersect(self, org, dirs, nthreads=1, tnear=1e-6):
        org, dirs = np.ascontiguousarray(org, np.float32), np.ascontiguousarray(dirs, np.float32)
        n = len(org)
        t, u, v, prim = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
        steps = (C.c_ulonglong * 2)()
        rc = load().simd_intersect(_p(self.nodes4), C.c_size_t(len(self.nodes4)), _p(self.slots), _p(org), _p(dirs), C.c_size_t(n), C.c_float(tnear), _p(t), _p(prim), _p(u), _p(v),
                                   C.c_int(nthreads), steps)
        if rc:
            raise RuntimeError("simd_intersect: traversal stack overflow")
        self.last_steps = (int(steps[0]), int(steps[1]))
        return t, prim, u, v

    def occluded(self, org, dirs, nthreads=1, tnear=1e-6):
        org, dirs = np.ascontiguousarray(org, np.float32), np.ascontiguousarray(dirs, np.float32)
        n = len(org)
        occ = np.zeros(n, np.uint8)
        steps = (C.c_ulonglong * 2)()
        rc = load().simd_occluded(_p(self.nodes4), C.c_size_t(len(self.nodes4)), _p(self.slots), _p(org), _p(dirs), C.c_size_t(n), C.c_float(tnear), _p(occ), C.c_int(nthreads), steps)
        if rc:
            raise RuntimeError("simd_occluded: traversal stack overflow")
        self.last_steps = (int(steps[0]), int(steps[1]))
        return occ.astype(bool)


_lib8 = None


def load8():
    """oracle/libsimd8_baseline.so (AVX2 + FMA, one ray against eight boxes per step); None where the host cannot run it"""
    global _lib8
    if _lib8 is None:
        path = os.path.join(_HERE, "libsimd8_baseline.so")
        if not os.path.exists(path):
            return None
        lib = C.CDLL(path)
        lib.simd8_supported.restype = C.c_int
        lib.simd8_build.restype = C.c_long
        lib.simd8_intersect.restype = C.c_int
        lib.simd8_occluded.restype = C.c_int
        _lib8 = lib
    return _lib8 if _lib8.simd8_supported() else None


class Tree8:
    """The same mesh walked 8-wide: the downloaded compressed 4-wide nodes collapsed once more on the host into 256-byte nodes of eight float boxes
    (simd8_build; not timed, like every build) and traversed one ray against eight boxes per step in AVX2 (oracle/simd8_baseline.c)."""

    def __init__(self, nodes4, slots):
        lib = load8()
        if lib is None:
            raise RuntimeError("the 8-wide baseline needs AVX2 + FMA on the host (and oracle/libsimd8_baseline.so)")
        nodes4 = np.ascontiguousarray(nodes4, np.uint32)
        self.slots = np.ascontiguousarray(slots, np.float32)
        raw = np.zeros(len(nodes4) * 64 + 16, np.int32)  # 256 bytes per node, at most one 8-wide node per 4-wide one; + room to align to 32 bytes
        off = (-raw.ctypes.data % 32) // 4
        self._raw = raw
        self.nodes8 = raw[off:off + len(nodes4) * 64]
        n8 = lib.simd8_build(_p(nodes4), C.c_size_t(len(nodes4)), _p(self.nodes8))
        if n8 < 0:
            raise RuntimeError("simd8_build failed (%d)" % n8)
        self.n8, self.n4 = int(n8), len(nodes4)
        kids = self.nodes8[:self.n8 * 64].reshape(-1, 64)[:, 56]
        self.children_per_node = float(kids.mean()) if self.n8 else 0.0

    def intersect(self, org, dirs, nthreads=1, tnear=1e-6):
        org, dirs = np.ascontiguousarray(org, np.float32), np.ascontiguousarray(dirs, np.float32)
        n = len(org)
        t, u, v, prim = np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.float32), np.zeros(n, np.int32)
        steps = (C.c_ulonglong * 2)()
        rc = load8().simd8_intersect(_p(self.nodes8), C.c_size_t(self.n8), _p(self.slots), _p(org), _p(dirs), C.c_size_t(n), C.c_float(tnear), _p(t), _p(prim), _p(u), _p(v),
                                     C.c_int(nthreads), steps)
        if rc:
            raise RuntimeError("simd8_intersect: traversal stack overflow")
        self.last_steps = (int(steps[0]), int(steps[1]))
        return t, prim, u, v

    def occluded(self, org, dirs, nthreads=1, tnear=1e-6):
        org, dirs = np.ascontiguousarray(org, np.float32), np.ascontiguousarray(dirs, np.float32)
        n = len(org)
        occ = np.zeros(n, np.uint8)
        steps = (C.c_ulonglong * 2)()
        rc = load8().simd8_occluded(_p(self.nodes8), C.c_size_t(self.n8), _p(self.slots), _p(org), _p(dirs), C.c_size_t(n), C.c_float(tnear), _p(occ), C.c_int(nthreads), steps)
        if rc:
            raise RuntimeError("simd8_occluded: traversal stack overflow")
        self.last_steps = (int(steps[0]), int(steps[1]))
        return occ.astype(bool)
