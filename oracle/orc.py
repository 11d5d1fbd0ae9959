"""ctypes binding of the CPU oracle (oracle/liboracle.so) and of the reference build (oracle/_ref).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg.  The product package (gravit_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")
REF_PATH = os.path.join(HERE, "_ref", "libgvtref.so")

RAY_DTYPE = np.dtype(
    {
        "names": ["origin", "t_min", "direction", "t_max", "color", "t", "id", "depth", "w", "type", "rng", "known"],
        "formats": [("<f4", 3), "<f4", ("<f4", 3), "<f4", ("<f4", 3), "<f4", "<i4", "<i4", "<f4", "<i4", "<u4", ("<u2", 6)],
        # rng: the per-ray RNG stream word in Ray::data[64..67]; known: the instances (+1) the ray has crossed without a hit on its current
        # segment, Ray::data[68..79] (the schedulers' known-miss shortcut) -- both unused by the reference
        "offsets": [0, 12, 16, 28, 32, 44, 48, 52, 56, 60, 64, 68],
        "itemsize": 80,
    }
)
MATERIAL_DTYPE = np.dtype(
    {
        "names": ["type", "ka", "ks", "kd", "alpha", "eta", "k", "roughness", "hsc", "backScattering", "hsFallOff"],
        "formats": ["<i4", ("<f4", 3), ("<f4", 3), ("<f4", 3), "<f4", ("<f4", 3), ("<f4", 3), "<f4", ("<f4", 3), "<f4", "<f4"],
        "offsets": [0, 4, 16, 28, 40, 44, 56, 68, 72, 84, 88],
        "itemsize": 92,
    }
)
LIGHT_DTYPE = np.dtype(
    {
        "names": ["type", "position", "color", "normal", "width", "height"],
        "formats": ["<i4", ("<f4", 3), ("<f4", 3), ("<f4", 3), "<f4", "<f4"],
        "offsets": [0, 4, 16, 28, 40, 44],
        "itemsize": 64,
    }
)
HIT_DTYPE = np.dtype([("t", "<f4"), ("prim", "<i4"), ("u", "<f4"), ("v", "<f4")])


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(os.path.join(HERE, "gvt_oracle.c")):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_mesh_create.restype = C.c_void_p
        _lib.orc_mesh_normals.restype = C.POINTER(C.c_float)
        _lib.orc_rng.restype = C.c_float
        _lib.orc_fastrand_lcg.restype = C.c_float
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None:
        a = a.reshape(shape)
    return a


def default_material():
    m = np.zeros(1, MATERIAL_DTYPE)
    m["type"] = 0
    m["kd"] = 0.5
    m["ks"] = 0.5
    m["alpha"] = 1.0
    m["eta"] = (0.19, 1.45, 1.50)
    m["k"] = (3.06, 2.40, 1.88)
    m["roughness"] = 0.05
    return m


def point_light(pos, color=(1, 1, 1)):
    l = np.zeros(1, LIGHT_DTYPE)
    l["type"] = 0
    l["position"] = pos
    l["color"] = color
    return l


class Mesh:
    def __init__(self, verts, tris, vnormals=None, vcolors=None, materials=None, face_mat=None, mesh_mat=None):
        self.verts = _f32(verts, (-1, 3))
        self.tris = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
        self.vnormals = None if vnormals is None else _f32(vnormals, (-1, 3))
        self.vcolors = None if vcolors is None else _f32(vcolors, (-1, 3))
        self.materials = None if materials is None else np.ascontiguousarray(materials, dtype=MATERIAL_DTYPE)
        self.face_mat = None if face_mat is None else np.ascontiguousarray(face_mat, dtype=np.int32)
        self.mesh_mat = None if mesh_mat is None else np.ascontiguousarray(mesh_mat, dtype=MATERIAL_DTYPE)
        self.h = C.c_void_p(
            lib().orc_mesh_create(
                _p(self.verts), C.c_size_t(len(self.verts)), _p(self.tris), C.c_size_t(len(self.tris)),
                _p(self.vnormals), _p(self.vcolors), _p(self.materials),
                C.c_size_t(0 if self.materials is None else len(self.materials)), _p(self.face_mat), _p(self.mesh_mat),
            )
        )

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_mesh_destroy(self.h)
            self.h = None

    def normals(self):
        p = lib().orc_mesh_normals(self.h)
        return np.ctypeslib.as_array(p, shape=(len(self.verts), 3)).copy()

    def bbox(self):
        lo = np.zeros(3, np.float32)
        hi = np.zeros(3, np.float32)
        lib().orc_mesh_bbox(self.h, _p(lo), _p(hi))
        return lo, hi

    def intersect(self, org, dirs, tnear=1e-6, use_bvh=True):
        org = _f32(org, (-1, 3))
        dirs = _f32(dirs, (-1, 3))
        out = np.zeros(len(org), HIT_DTYPE)
        lib().orc_intersect(self.h, _p(org), _p(dirs), C.c_size_t(len(org)), C.c_float(tnear), C.c_int(int(use_bvh)), _p(out))
        return out

    def occluded(self, org, dirs, tnear=1e-6, use_bvh=True):
        org = _f32(org, (-1, 3))
        dirs = _f32(dirs, (-1, 3))
        out = np.zeros(len(org), np.int32)
        lib().orc_occluded(self.h, _p(org), _p(dirs), C.c_size_t(len(org)), C.c_float(tnear), C.c_int(int(use_bvh)), _p(out))
        return out

    def trace(self, rays, m, minv, normi, lights, normal_mode=0, seed=0, nthreads=1, begin=0, end=0):
        """Adapter::trace.  `rays` (RAY_DTYPE) is updated in place like the reference's rayList."""
        assert rays.dtype == RAY_DTYPE and rays.flags.c_contiguous
        if end == 0:
            end = len(rays)
        lights = np.ascontiguousarray(lights, dtype=LIGHT_DTYPE)
        cap = max(16, (end - begin) * (1 + len(lights)))
        out = np.zeros(cap, RAY_DTYPE)
        n = C.c_size_t(0)
        rc = lib().orc_trace(
            self.h, _p(rays), C.c_size_t(begin), C.c_size_t(end), _p(out), C.c_size_t(cap), C.byref(n),
            _p(_f32(m, 16)), _p(_f32(minv, 16)), _p(_f32(normi, 9)), _p(lights), C.c_size_t(len(lights)),
            C.c_int(normal_mode), C.c_uint32(seed), C.c_int(nthreads),
        )
        assert rc == 0
        return out[: n.value].copy()


def trace_counts():
    a, b = C.c_uint64(0), C.c_uint64(0)
    lib().orc_trace_counts(C.byref(a), C.byref(b))
    return a.value, b.value


def generate_normals(verts, tris):
    verts = _f32(verts, (-1, 3))
    tris = np.ascontiguousarray(tris, dtype=np.int32).reshape(-1, 3)
    out = np.zeros_like(verts)
    lib().orc_generate_normals(_p(verts), C.c_size_t(len(verts)), _p(tris), C.c_size_t(len(tris)), _p(out))
    return out


def shade(mat, ray, N, light, light_pos):
    c = np.zeros(3, np.float32)
    ok = lib().orc_shade(_p(np.ascontiguousarray(mat, MATERIAL_DTYPE)), _p(np.ascontiguousarray(ray, RAY_DTYPE)), _p(_f32(N, 3)),
                         _p(np.ascontiguousarray(light, LIGHT_DTYPE)), _p(_f32(light_pos, 3)), _p(c))
    return bool(ok), c


def rng(seed):
    s = C.c_uint32(seed)
    v = lib().orc_rng(C.byref(s))
    return v, s.value


def fastrand_lcg(seed, mn=0.0, mx=1.0):
    s = C.c_uint32(seed)
    v = lib().orc_fastrand_lcg(C.byref(s), C.c_float(mn), C.c_float(mx))
    return v, s.value


def cos_weighted_dir(n, seed):
    s = C.c_uint32(seed)
    out = np.zeros(3, np.float32)
    lib().orc_cos_weighted_dir(_p(_f32(n, 3)), C.byref(s), _p(out))
    return out, s.value


def math_probe(kind, x):
    """include/gvt_math.h on the host (kind 0 sinf, 1 cosf, 2 (float)acos(sqrt(1-x))); kind + 16: the libm calls of the reference."""
    x = _f32(x, -1)
    out = np.zeros_like(x)
    lib().orc_math_probe(C.c_int(kind), _p(x), C.c_size_t(len(x)), _p(out))
    return out


def camera_rays(eye, focus, up, fov, width, height, samples=1, depth=1, jitter=0.0):
    rays = np.zeros(width * height * samples * samples, RAY_DTYPE)
    lib().orc_camera_generate(_p(_f32(eye, 3)), _p(_f32(focus, 3)), _p(_f32(up, 3)), C.c_float(fov), C.c_int(width),
                              C.c_int(height), C.c_int(samples), C.c_int(depth), C.c_float(jitter), _p(rays))
    return rays


def toplevel_order(inst_lo, inst_hi):
    lo = _f32(inst_lo, (-1, 3))
    hi = _f32(inst_hi, (-1, 3))
    order = np.zeros(len(lo), np.int32)
    lib().orc_toplevel_order(_p(lo), _p(hi), C.c_size_t(len(lo)), _p(order))
    return order


def toplevel_intersect(inst_lo, inst_hi, order, rays, frm=-1):
    lo = _f32(inst_lo, (-1, 3))
    hi = _f32(inst_hi, (-1, 3))
    order = np.ascontiguousarray(order, np.int32)
    nxt = np.zeros(len(rays), np.int32)
    t = np.zeros(len(rays), np.float32)
    lib().orc_toplevel_intersect(_p(lo), _p(hi), _p(order), C.c_size_t(len(lo)), _p(rays), C.c_size_t(len(rays)),
                                 C.c_int(frm), _p(nxt), _p(t))
    return nxt, t


def set_skip_known_misses(on):
    """The build's image-identical shortcut of shuffleRays (gvt_oracle.c "known misses"); off = the reference's hop-by-hop behaviour."""
    lib().orc_set_skip_known_misses(C.c_int(1 if on else 0))


def get_skip_known_misses():
    return bool(lib().orc_get_skip_known_misses())


def shuffle_step(inst_lo, inst_hi, order, rays, frm=-1):
    """shuffleRays' decision per ray, in place (origins advanced, known-miss lists updated): the next instance or -1."""
    lo, hi = _f32(inst_lo, (-1, 3)), _f32(inst_hi, (-1, 3))
    order = np.ascontiguousarray(order, np.int32)
    nxt = np.zeros(len(rays), np.int32)
    lib().orc_shuffle_step(_p(lo), _p(hi), _p(order), C.c_size_t(len(lo)), _p(rays), C.c_size_t(len(rays)), C.c_int(frm), _p(nxt))
    return nxt


class _Scene(C.Structure):
    _fields_ = [
        ("meshes", C.POINTER(C.c_void_p)), ("m", C.c_void_p), ("minv", C.c_void_p), ("normi", C.c_void_p),
        ("inst_lo", C.c_void_p), ("inst_hi", C.c_void_p), ("nInst", C.c_size_t), ("lights", C.c_void_p),
        ("nLights", C.c_size_t), ("normal_mode", C.c_int), ("nthreads", C.c_int),
    ]


class FrameStats(C.Structure):
    _fields_ = [("rays_closest", C.c_uint64), ("rays_any", C.c_uint64), ("adapter_calls", C.c_uint64),
                ("rays_sent", C.c_uint64), ("rounds", C.c_uint64)]


def _scene_struct(meshes, m, minv, normi, inst_lo, inst_hi, lights, normal_mode, nthreads):
    keep = dict(
        arr=(C.c_void_p * len(meshes))(*[x.h for x in meshes]), m=_f32(m, (-1, 16)), minv=_f32(minv, (-1, 16)),
        normi=_f32(normi, (-1, 9)), lo=_f32(inst_lo, (-1, 3)), hi=_f32(inst_hi, (-1, 3)),
        lights=np.ascontiguousarray(lights, LIGHT_DTYPE),
    )
    s = _Scene(keep["arr"], _p(keep["m"]), _p(keep["minv"]), _p(keep["normi"]), _p(keep["lo"]), _p(keep["hi"]),
               len(meshes), _p(keep["lights"]), len(keep["lights"]), normal_mode, nthreads)
    return s, keep


def render_image(meshes, m, minv, normi, inst_lo, inst_hi, lights, cam_rays, width, height, normal_mode=0, nthreads=1):
    s, keep = _scene_struct(meshes, m, minv, normi, inst_lo, inst_hi, lights, normal_mode, nthreads)
    fb = np.zeros((height, width, 4), np.float32)
    rays = cam_rays.copy()
    st = FrameStats()
    lib().orc_render_image(C.byref(s), _p(rays), C.c_size_t(len(rays)), C.c_int(width), C.c_int(height), _p(fb), C.byref(st))
    return fb, st


def render_domain(meshes, m, minv, normi, inst_lo, inst_hi, lights, owner, P, cam_rays, width, height, normal_mode=0, nthreads=1):
    s, keep = _scene_struct(meshes, m, minv, normi, inst_lo, inst_hi, lights, normal_mode, nthreads)
    fb = np.zeros((height, width, 4), np.float32)
    owner = np.ascontiguousarray(owner, np.int32)
    st = FrameStats()
    lib().orc_render_domain(C.byref(s), _p(owner), C.c_int(P), _p(cam_rays), C.c_size_t(len(cam_rays)), C.c_int(width),
                            C.c_int(height), _p(fb), C.byref(st))
    return fb, st


def fb_to_ppm_bytes(fb):
    h, w = fb.shape[:2]
    out = np.zeros(h * w * 3, np.uint8)
    lib().orc_fb_to_ppm_bytes(_p(np.ascontiguousarray(fb, np.float32)), C.c_int(w), C.c_int(h), _p(out))
    return out.reshape(h, w, 3)


# ---------------------------------------------------------------- reference build (oracle/_ref)
_ref = None


def ref():
    """The reference's own Ray/Material/Mesh/BBox/Light code (oracle/_ref/libgvtref.so), or None."""
    global _ref
    if _ref is None and os.path.exists(REF_PATH):
        _ref = C.CDLL(REF_PATH)
        _ref.ref_rng.restype = C.c_float
        _ref.ref_fastrand_lcg.restype = C.c_float
        _ref.ref_ray_epsilon.restype = C.c_float
        _ref.ref_box_surface_area.restype = C.c_float
    return _ref
