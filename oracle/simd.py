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
