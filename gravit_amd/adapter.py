"""Host-side mirror of GraviT's engine-adapter interface for the gfx950 adapter.

  gvt::render::Adapter                      src/gvt/render/Adapter.h:44-88
  gvt::render::adapter::embree::data::EmbreeMeshAdapter   adapter/embree/EmbreeMeshAdapter.{h,cpp}

`HipMeshAdapter` has the reference's shape: constructed from a Mesh, one `trace()` with the reference's
arguments (rayList, m, minv, normi, lights, begin, end) and the reference's output contract (moved
rays = misses + un-occluded shadow rays, rayList updated in place).  Everything runs in
libgvt_hip.so; a missing library or device raises (no CPU path).
"""
import ctypes as C

import numpy as np

from . import capi
from .layouts import HIT_DTYPE, LIGHT_DTYPE, MATERIAL_DTYPE, NORMALS_FLAT, RAY_DTYPE


class RayQueue:
    """Device-resident gvt::render::actor::RayVector (actor/Ray.h:189)."""

    def __init__(self, capacity=0):
        self.lib = capi.load()
        self.h = C.c_void_p(self.lib.gvt_hip_queue_create(C.c_size_t(capacity)))
        if not self.h:
            raise capi.GvtHipError("gvt_hip_queue_create: " + capi.last_error())

    def close(self):
        if getattr(self, "h", None):
            self.lib.gvt_hip_queue_destroy(self.h)
            self.h = None

    __del__ = close

    def __len__(self):
        n = C.c_size_t(0)
        capi.check(self.lib.gvt_hip_queue_size(self.h, C.byref(n)), "gvt_hip_queue_size")
        return n.value

    def clear(self):
        capi.check(self.lib.gvt_hip_queue_clear(self.h), "gvt_hip_queue_clear")

    def reserve(self, n):
        capi.check(self.lib.gvt_hip_queue_reserve(self.h, C.c_size_t(n)), "gvt_hip_queue_reserve")

    def append(self, rays, keep_state=False):
        """host rays; keep_state: they are rays this library exported (bytes 64..79 = stream word + known misses), else they start fresh"""
        rays = np.ascontiguousarray(rays, dtype=RAY_DTYPE)
        capi.check(self.lib.gvt_hip_queue_append_flags(self.h, capi.ptr(rays), C.c_size_t(len(rays)), C.c_int(2 if keep_state else 0)), "gvt_hip_queue_append_flags")

    def append_device(self, dptr, n, keep_state=True):
        """n 80-byte rays at device address dptr (a received wire buffer: the rays keep their state)."""
        capi.check(self.lib.gvt_hip_queue_append_flags(self.h, C.c_void_p(dptr), C.c_size_t(n), C.c_int(1 | (2 if keep_state else 0))), "gvt_hip_queue_append_flags")

    def export_device(self, dptr, cap):
        n = C.c_size_t(0)
        capi.check(self.lib.gvt_hip_queue_export(self.h, C.c_void_p(dptr), C.c_size_t(cap), C.byref(n), C.c_int(1)), "gvt_hip_queue_export")
        return n.value

    def to_numpy(self):
        n = len(self)
        out = np.zeros(n, RAY_DTYPE)
        got = C.c_size_t(0)
        capi.check(self.lib.gvt_hip_queue_export(self.h, capi.ptr(out), C.c_size_t(n), C.byref(got), C.c_int(0)), "gvt_hip_queue_export")
        return out


class HipMeshAdapter:
    """gvt::render::adapter::hip::data::HipMeshAdapter -- drop-in for EmbreeMeshAdapter.

    normal_mode: NORMALS_FLAT is the current EmbreeMeshAdapter.cpp (FLAT_SHADING, :75); NORMALS_SMOOTH is
    what the reference's golden images, EmbreeStreamMeshAdapter and the OptiX adapter use.
    """

    def __init__(self, mesh, normal_mode=NORMALS_FLAT):
        self.lib = capi.load()
        self.normal_mode = int(normal_mode)
        self.verts = capi.f32(mesh.verts, (-1, 3))
        self.tris = np.ascontiguousarray(mesh.tris, dtype=np.int32).reshape(-1, 3)
        vn = None if mesh.vnormals is None else capi.f32(mesh.vnormals, (-1, 3))
        vc = None if mesh.vcolors is None else capi.f32(mesh.vcolors, (-1, 3))
        mats = None if mesh.materials is None else np.ascontiguousarray(mesh.materials, dtype=MATERIAL_DTYPE)
        fm = None if mesh.face_mat is None else np.ascontiguousarray(mesh.face_mat, dtype=np.int32)
        mm = None if mesh.material is None else np.ascontiguousarray(mesh.material, dtype=MATERIAL_DTYPE)
        self.h = C.c_void_p(self.lib.gvt_hip_mesh_create(
            capi.ptr(self.verts), C.c_size_t(len(self.verts)), capi.ptr(self.tris), C.c_size_t(len(self.tris)), capi.ptr(vn),
            capi.ptr(vc), capi.ptr(mats), C.c_size_t(0 if mats is None else len(mats)), capi.ptr(fm), capi.ptr(mm)))
        if not self.h:
            raise capi.GvtHipError("gvt_hip_mesh_create: " + capi.last_error())

    def close(self):
        if getattr(self, "h", None):
            self.lib.gvt_hip_mesh_destroy(self.h)
            self.h = None

    __del__ = close

    def info(self):
        i = capi.MeshInfo()
        capi.check(self.lib.gvt_hip_mesh_get_info(self.h, C.byref(i)), "gvt_hip_mesh_get_info")
        return {"n_tris": i.n_tris, "n_verts": i.n_verts, "n_nodes": i.n_nodes, "n_leaves": i.n_leaves, "bbox_lo": list(i.bbox_lo),
                "bbox_hi": list(i.bbox_hi), "build_ms": i.build_ms, "max_leaf": i.max_leaf, "bytes_nodes": i.bytes_nodes, "bytes_tris": i.bytes_tris,
                "packet": int(i.packet), "sah_inner": float(i.sah_inner)}

    def normals(self):
        out = np.zeros((len(self.verts), 3), np.float32)
        capi.check(self.lib.gvt_hip_mesh_get_normals(self.h, capi.ptr(out)), "gvt_hip_mesh_get_normals")
        return out

    # -- Adapter::trace (Adapter.h:82-84) ------------------------------------------------------------
    def trace(self, rayList, m, minv, normi, lights, begin=0, end=0, seed=0, write_back=True, out=None):
        """Traces rayList[begin:end) (end==0 -> all, EmbreeMeshAdapter.cpp:642).  rayList (RAY_DTYPE, C-contiguous)
        is updated in place (write_back=False: only read, GVT_HIP_TRACE_NO_WRITEBACK); returns moved_rays (misses + un-occluded
        shadow rays, order unspecified), a view of `out` when the caller brings its own buffer (like a re-used RayVector)."""
        if rayList.dtype != RAY_DTYPE or not rayList.flags.c_contiguous:
            raise ValueError("rayList must be a C-contiguous array of RAY_DTYPE (80-byte gvt Ray)")
        lights = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        n = len(rayList)
        e = n if end == 0 else end
        cap = max(16, (e - begin) * (1 + len(lights)))
        if out is None:
            out = np.zeros(cap, RAY_DTYPE)
        elif out.dtype != RAY_DTYPE or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous array of RAY_DTYPE")
        cap = len(out)
        n_out = C.c_size_t(0)
        capi.check(self.lib.gvt_hip_trace_ex(
            self.h, capi.ptr(rayList), C.c_size_t(n), C.c_size_t(begin), C.c_size_t(end), capi.ptr(out), C.c_size_t(cap),
            C.byref(n_out), capi.ptr(capi.f32(m, 16)), capi.ptr(capi.f32(minv, 16)), capi.ptr(capi.f32(normi, 9)), capi.ptr(lights),
            C.c_size_t(len(lights)), C.c_int(self.normal_mode), C.c_uint32(seed), C.c_uint32(0 if write_back else 1)), "gvt_hip_trace_ex")
        return out[: n_out.value]

    def trace_queue(self, q_in, q_out, m, minv, normi, lights, seed=0, sink=None):
        """Adapter::trace on device-resident queues: q_in is consumed, moved rays are appended to q_out.  sink=(top, from_inst, fb):
        un-occluded shadow rays that meet no other instance deposit into fb inside the adapter (gvt_hip_trace_queue_sink)."""
        lights = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        top, from_inst, fb = sink if sink is not None else (None, -1, None)
        capi.check(self.lib.gvt_hip_trace_queue_sink(
            self.h, q_in.h, q_out.h, capi.ptr(capi.f32(m, 16)), capi.ptr(capi.f32(minv, 16)), capi.ptr(capi.f32(normi, 9)),
            capi.ptr(lights), C.c_size_t(len(lights)), C.c_int(self.normal_mode), C.c_uint32(seed),
            top.h if top is not None else None, C.c_int(from_inst), fb.h if fb is not None else None), "gvt_hip_trace_queue_sink")

    # -- the two Embree queries underneath (EmbreeMeshAdapter.cpp:474,375) -------------------------------
    def intersect(self, org, dirs, tnear=1e-6):
        org = capi.f32(org, (-1, 3))
        dirs = capi.f32(dirs, (-1, 3))
        out = np.zeros(len(org), HIT_DTYPE)
        capi.check(self.lib.gvt_hip_intersect(self.h, capi.ptr(org), capi.ptr(dirs), C.c_size_t(len(org)), C.c_float(tnear), capi.ptr(out)),
                   "gvt_hip_intersect")
        return out

    def wide_visit_stats(self, org, dirs, width, tnear=1e-6):
        """Diagnostic: nodes a `width`-wide collapse of the tree would make each ray visit (mean), and the collapse's node count."""
        org = capi.f32(org, (-1, 3))
        dirs = capi.f32(dirs, (-1, 3))
        cnt = np.zeros(len(org), np.uint32)
        nw = C.c_uint64(0)
        capi.check(self.lib.gvt_hip_wide_visit_stats(self.h, capi.ptr(org), capi.ptr(dirs), C.c_size_t(len(org)), C.c_float(tnear), C.c_int(width),
                                                     capi.ptr(cnt), C.byref(nw)), "gvt_hip_wide_visit_stats")
        return {"width": width, "nodes_per_ray": float(cnt.mean()), "p99": float(np.percentile(cnt, 99)), "max": int(cnt.max()), "wide_nodes": int(nw.value)}

    def download_nodes(self):
        """Diagnostic: the binary LBVH as built, (n_nodes, 16) float32 rows (gvt_hip.h gvt_hip_mesh_download_nodes; child refs bit-cast in columns 12, 13)."""
        n = self.info()["n_nodes"]
        out = np.zeros((n, 16), np.float32)
        capi.check(self.lib.gvt_hip_mesh_download_nodes(self.h, capi.ptr(out), C.c_size_t(n)), "gvt_hip_mesh_download_nodes")
        return out

    def download_wide(self):
        """Measurement: the traversal layout -- (n4, 16) uint32 compressed 4-wide nodes and (n_tris, 16) float32 triangle slots in leaf order."""
        i = self.info()
        n4 = i["bytes_nodes"] // 64 - i["n_nodes"]
        nodes4 = np.zeros((n4, 16), np.uint32)
        slots = np.zeros((i["n_tris"], 16), np.float32)
        capi.check(self.lib.gvt_hip_mesh_download_wide(self.h, capi.ptr(nodes4), C.c_size_t(n4), capi.ptr(slots), C.c_size_t(i["n_tris"])), "gvt_hip_mesh_download_wide")
        return nodes4, slots

    def download_clusters(self):
        """Measurement: the cluster layout of the 4-wide nodes (built on first use) -- (n4, 16) uint32 and the root's entry, or (None, -1) when the mesh has none."""
        i = self.info()
        n4 = i["bytes_nodes"] // 64 - i["n_nodes"]
        out = np.zeros((n4, 16), np.uint32)
        root = C.c_int32(-1)
        capi.check(self.lib.gvt_hip_mesh_download_clusters(self.h, capi.ptr(out), C.c_size_t(n4), C.byref(root)), "gvt_hip_mesh_download_clusters")
        return (out, root.value) if root.value >= 0 else (None, -1)

    def upload_nodes(self, nodes):
        """Diagnostic: replace the binary nodes (visit-count diagnostics only) by a tree over the same leaves."""
        nodes = np.ascontiguousarray(nodes, np.float32)
        capi.check(self.lib.gvt_hip_mesh_upload_nodes(self.h, capi.ptr(nodes), C.c_size_t(len(nodes))), "gvt_hip_mesh_upload_nodes")

    def marked_visit_stats(self, org, dirs, marks, tnear=1e-6):
        """Diagnostic: per ray, the marked binary nodes (roots of the wide nodes of a collapse the caller chose) its closest-hit traversal visits."""
        org = capi.f32(org, (-1, 3))
        dirs = capi.f32(dirs, (-1, 3))
        marks = np.ascontiguousarray(marks, np.uint8)
        cnt = np.zeros(len(org), np.uint32)
        capi.check(self.lib.gvt_hip_marked_visit_stats(self.h, capi.ptr(org), capi.ptr(dirs), C.c_size_t(len(org)), C.c_float(tnear), capi.ptr(marks), capi.ptr(cnt)),
                   "gvt_hip_marked_visit_stats")
        return cnt

    def visit_stats(self, org, dirs, tnear=1e-6):
        """Diagnostic: per-ray (inner-node visits, leaf visits, triangle tests) of the closest-hit traversal, plus the
        number of steps a 64-lane wave executes per batch (the slowest lane's inner + leaf steps)."""
        org = capi.f32(org, (-1, 3))
        dirs = capi.f32(dirs, (-1, 3))
        n = len(org)
        cnt = np.zeros((n, 3), np.uint32)
        capi.check(self.lib.gvt_hip_visit_stats(self.h, capi.ptr(org), capi.ptr(dirs), C.c_size_t(n), C.c_float(tnear), capi.ptr(cnt)),
                   "gvt_hip_visit_stats")
        steps = (cnt[:, 0] + cnt[:, 1]).astype(np.int64)
        pad = (-n) % 64
        waves = np.concatenate([steps, np.zeros(pad, np.int64)]).reshape(-1, 64)
        return {"inner_per_ray": float(cnt[:, 0].mean()), "leaf_per_ray": float(cnt[:, 1].mean()), "tri_tests_per_ray": float(cnt[:, 2].mean()),
                "lane_steps_per_ray": float(steps.mean()), "wave_steps_per_batch": float(waves.max(axis=1).mean()),
                "simd_efficiency": float(steps.sum() / max(1, waves.max(axis=1).sum() * 64)), "counts": cnt}

    def occluded(self, org, dirs, tnear=1e-6):
        org = capi.f32(org, (-1, 3))
        dirs = capi.f32(dirs, (-1, 3))
        out = np.zeros(len(org), np.int32)
        capi.check(self.lib.gvt_hip_occluded(self.h, capi.ptr(org), capi.ptr(dirs), C.c_size_t(len(org)), C.c_float(tnear), capi.ptr(out)),
                   "gvt_hip_occluded")
        return out


class TopLevel:
    """Top-level instance set: gvt::render::data::accel::BVH (accel/BVH.h) + shuffleRays (TracerBase.h:325-414)."""

    def __init__(self, inst_lo, inst_hi):
        self.lib = capi.load()
        self.lo = capi.f32(inst_lo, (-1, 3))
        self.hi = capi.f32(inst_hi, (-1, 3))
        self.n = len(self.lo)
        self.h = C.c_void_p(self.lib.gvt_hip_top_create(capi.ptr(self.lo), capi.ptr(self.hi), C.c_size_t(self.n)))
        if not self.h:
            raise capi.GvtHipError("gvt_hip_top_create: " + capi.last_error())

    def close(self):
        if getattr(self, "h", None):
            self.lib.gvt_hip_top_destroy(self.h)
            self.h = None

    __del__ = close

    def order(self):
        out = np.zeros(self.n, np.int32)
        capi.check(self.lib.gvt_hip_top_order(self.h, capi.ptr(out)), "gvt_hip_top_order")
        return out

    def shuffle(self, q_in, from_inst, queues, fb, keep_mask=None):
        arr = (C.c_void_p * max(1, self.n))(*[q.h for q in queues])
        km = None if keep_mask is None else np.ascontiguousarray(keep_mask, dtype=np.uint8)
        capi.check(self.lib.gvt_hip_shuffle(self.h, q_in.h, C.c_int(from_inst), arr, capi.ptr(km), fb.h if fb is not None else None),
                   "gvt_hip_shuffle")


class FrameBuffer:
    """Float RGBA framebuffer: gvt::render::composite::IceTComposite (composite/IceTComposite.cpp:79-157)."""

    def __init__(self, width, height):
        self.lib = capi.load()
        self.w, self.hgt = width, height
        self.h = C.c_void_p(self.lib.gvt_hip_fb_create(C.c_int(width), C.c_int(height)))
        if not self.h:
            raise capi.GvtHipError("gvt_hip_fb_create: " + capi.last_error())

    def close(self):
        if getattr(self, "h", None):
            self.lib.gvt_hip_fb_destroy(self.h)
            self.h = None

    __del__ = close

    def clear(self):
        capi.check(self.lib.gvt_hip_fb_clear(self.h), "gvt_hip_fb_clear")

    def device_ptr(self):
        return self.lib.gvt_hip_fb_device_ptr(self.h)

    def download(self, clamp=True):
        out = np.zeros((self.hgt, self.w, 4), np.float32)
        capi.check(self.lib.gvt_hip_fb_download(self.h, capi.ptr(out), C.c_int(int(clamp))), "gvt_hip_fb_download")
        return out

    def ppm_bytes(self):
        out = np.zeros((self.hgt, self.w, 3), np.uint8)
        capi.check(self.lib.gvt_hip_fb_write_ppm_bytes(self.h, capi.ptr(out)), "gvt_hip_fb_write_ppm_bytes")
        return out


def camera_generate(q, cam, tile=0):
    """gvtPerspectiveCamera::generateRays into a device queue; tile=8 lists the same rays in 8x8-pixel tiles."""
    capi.check(capi.load().gvt_hip_camera_generate_tiled(
        q.h, capi.ptr(capi.f32(cam.eye, 3)), capi.ptr(capi.f32(cam.focus, 3)), capi.ptr(capi.f32(cam.up, 3)), C.c_float(cam.fov),
        C.c_int(cam.width), C.c_int(cam.height), C.c_int(cam.samples), C.c_int(cam.depth), C.c_float(cam.jitter), C.c_int(tile)),
        "gvt_hip_camera_generate_tiled")
