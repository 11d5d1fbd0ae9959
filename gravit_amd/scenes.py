"""Scene inputs for the adapter path: the reference's two golden scenes, the BASELINE.json
synthetic configurations, and tiny OBJ / PLY readers.  Pure numpy, no device code.

Citations are relative to the GraviT tree (/root/reference in the build container).
"""
import os
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from .layouts import LIGHT_DTYPE, MATERIAL_DTYPE, default_material, point_light

F = np.float32


@dataclass
class MeshData:
    """gvt::render::data::primitives::Mesh inputs (Mesh.h:87-100): 0-based faces."""

    verts: np.ndarray  # (nV,3) f32
    tris: np.ndarray  # (nT,3) i32
    material: np.ndarray = field(default_factory=default_material)  # Mesh::mat
    vnormals: Optional[np.ndarray] = None
    vcolors: Optional[np.ndarray] = None
    materials: Optional[np.ndarray] = None  # faces_to_materials table
    face_mat: Optional[np.ndarray] = None

    def bbox(self):
        v = self.verts  # per column: a min over axis 0 of an (n, 3) array walks three-element rows (1.5 s at 30 M vertices; this: 0.1 s)
        return np.array([v[:, k].min() for k in range(3)], F), np.array([v[:, k].max() for k in range(3)], F)


@dataclass
class Camera:
    """gvtPerspectiveCamera parameters (gvtCamera.cpp:233-312); fov in radians."""

    eye: tuple
    focus: tuple
    up: tuple
    fov: float
    width: int
    height: int
    samples: int = 1
    depth: int = 1
    jitter: float = 0.0


@dataclass
class Scene:
    meshes: List[MeshData]
    inst_mesh: List[int]  # instance -> mesh index
    m: np.ndarray  # (nInst,16) glm column-major
    minv: np.ndarray  # (nInst,16)
    normi: np.ndarray  # (nInst,9)
    inst_lo: np.ndarray  # (nInst,3) world AABB (api.cpp:309-312)
    inst_hi: np.ndarray
    lights: np.ndarray  # LIGHT_DTYPE
    camera: Camera
    name: str = ""

    @property
    def n_inst(self):
        return len(self.inst_mesh)


# ------------------------------------------------------------------ matrices (glm, column-major)
def mat_translate_scale(t, s):
    """glm::scale(glm::translate(I, t), s) as 16 floats, column-major (SimpleApp.cpp:166-168)."""
    m = np.zeros((4, 4), F)  # m[col][row]
    m[0, 0], m[1, 1], m[2, 2] = F(s[0]), F(s[1]), F(s[2])
    m[3, 0], m[3, 1], m[3, 2], m[3, 3] = F(t[0]), F(t[1]), F(t[2]), F(1)
    return m.reshape(16)


def instance_matrices(m16):
    """minv = inverse(m), normi = transpose(inverse(mat3(m)))  (api.cpp:307-308)."""
    M = m16.reshape(4, 4).T.astype(np.float64)  # row-major math matrix
    Minv = np.linalg.inv(M)
    minv = Minv.T.astype(F).reshape(16)
    N = np.linalg.inv(M[:3, :3]).T  # math matrix of normi
    normi = N.T.astype(F).reshape(9)  # column-major storage
    return minv, normi


def instance_bbox(m16, lo, hi):
    """Box3D(M*min, M*max): only the two corners are transformed (api.cpp:309-312)."""
    M = m16.reshape(4, 4).T
    a = (M @ np.array([lo[0], lo[1], lo[2], 1], F)).astype(F)[:3]
    b = (M @ np.array([hi[0], hi[1], hi[2], 1], F)).astype(F)[:3]
    return np.minimum(a, b), np.maximum(a, b)


def add_faces_1based(verts, faces1):
    """Mesh::addFace (Mesh.cpp:102-112): 1-based indices, faces with coincident vertices are dropped."""
    out = []
    for a, b, c in np.asarray(faces1).reshape(-1, 3):
        va, vb, vc = verts[a - 1], verts[b - 1], verts[c - 1]
        if (va == vb).all() or (vb == vc).all() or (vc == va).all():
            continue
        out.append((a - 1, b - 1, c - 1))
    return np.array(out, np.int32).reshape(-1, 3)


def _assemble(meshes, inst_mesh, mats, lights, camera, name):
    minv, normi, lo, hi = [], [], [], []
    for i, m in zip(inst_mesh, mats):
        a, b = instance_matrices(m)
        minv.append(a)
        normi.append(b)
        l, h = instance_bbox(m, *meshes[i].bbox())
        lo.append(l)
        hi.append(h)
    return Scene(meshes, list(inst_mesh), np.array(mats, F).reshape(-1, 16), np.array(minv, F).reshape(-1, 16),
                 np.array(normi, F).reshape(-1, 9), np.array(lo, F).reshape(-1, 3), np.array(hi, F).reshape(-1, 3),
                 np.ascontiguousarray(lights, LIGHT_DTYPE), camera, name)


# ------------------------------------------------------------------ golden scene 1: gvtSimple
def simple_scene(width=512, height=512):
    """The 5x5 cone/cube grid of src/apps/render/SimpleApp.cpp:82-233 (golden simple.ppm)."""
    cone_v = np.array([0.5, 0.0, 0.0, -0.5, 0.5, 0.0, -0.5, 0.25, 0.433013, -0.5, -0.25, 0.43013, -0.5, -0.5, 0.0, -0.5, -0.25,
                       -0.433013, -0.5, 0.25, -0.433013], F).reshape(-1, 3)
    cone_f = [1, 2, 3, 1, 3, 4, 1, 4, 5, 1, 5, 6, 1, 6, 7, 1, 7, 2]
    cube_v = np.array([-0.5, -0.5, 0.5, 0.5, -0.5, 0.5, 0.5, 0.5, 0.5, -0.5, 0.5, 0.5, -0.5, -0.5, -0.5, 0.5, -0.5, -0.5, 0.5, 0.5,
                       -0.5, -0.5, 0.5, -0.5, 0.5, 0.5, 0.5, -0.5, 0.5, 0.5, 0.5, 0.5, -0.5, -0.5, 0.5, -0.5, -0.5, -0.5, 0.5, 0.5,
                       -0.5, 0.5, -0.5, -0.5, -0.5, 0.5, -0.5, -0.5, 0.5, -0.5, 0.5, 0.5, 0.5, 0.5, 0.5, -0.5, -0.5, 0.5, 0.5, -0.5,
                       -0.5, -0.5, 0.5, -0.5, 0.5, 0.5, -0.5, -0.5, -0.5, -0.5, 0.5, -0.5], F).reshape(-1, 3)
    cube_f = [1, 2, 3, 1, 3, 4, 17, 19, 20, 17, 20, 18, 6, 5, 8, 6, 8, 7, 23, 21, 22, 23, 22, 24, 10, 9, 11, 10, 11, 12, 13, 15,
              16, 13, 16, 14]
    white = default_material(kd=(1.0, 1.0, 1.0))  # addMeshMaterial(LAMBERT, kd=1, alpha=1) api.cpp:228-236
    cone = MeshData(cone_v, add_faces_1based(cone_v, cone_f), white)
    cube = MeshData(cube_v, add_faces_1based(cube_v, cube_f), white)
    mats, inst_mesh = [], []
    inst = 0
    for i in range(-2, 3):
        for j in range(-2, 3):
            mats.append(mat_translate_scale((0.0, i * 0.5, j * 0.5), (0.4, 0.4, 0.4)))
            inst_mesh.append(1 if inst % 2 else 0)
            inst += 1
    cam = Camera((4.0, 0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), float(F(45.0 * np.pi / 180.0)), width, height, 1, 1, 0.5)
    return _assemble([cone, cube], inst_mesh, mats, point_light((1.0, 0.0, -1.0)), cam, "simple")


# ------------------------------------------------------------------ readers
def read_obj(path):
    """Vertices and triangles of a Wavefront OBJ (faces 0-based, no degenerate filter, like
    data/reader/ObjReader.cpp:122-135 which pushes tinyobj faces straight into Mesh::faces)."""
    vs, fs = [], []
    with open(path) as f:
        for line in f:
            if line.startswith("v "):
                p = line.split()
                vs.append((float(p[1]), float(p[2]), float(p[3])))
            elif line.startswith("f "):
                idx = [int(tok.split("/")[0]) for tok in line.split()[1:]]
                idx = [i - 1 if i > 0 else len(vs) + i for i in idx]
                for k in range(1, len(idx) - 1):  # fan-triangulate like tinyobj
                    fs.append((idx[0], idx[k], idx[k + 1]))
    return np.array(vs, F).reshape(-1, 3), np.array(fs, np.int32).reshape(-1, 3)


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def read_ply_full(path):
    """PLY as PlyReader.cpp:54-176 takes it through ply.c: `ascii`, `binary_little_endian` or `binary_big_endian`; a `vertex` element whose
    x y z become the positions and -- when it has more than five properties (PlyReader.cpp:123) -- whose red green blue (uchar) become vertex
    colours / 255; a `face` element with one list property of 3 indices (PlyReader.cpp:163-167 copies verts[0..2]).  Other elements
    are skipped.  Returns (verts f32 [nv,3], tris i32 [nf,3], colors f32 [nv,3] or None)."""
    with open(path, "rb") as f:
        if f.readline().strip() != b"ply":
            raise ValueError("%s: not a PLY file" % path)
        fmt, elems = None, []  # elems: [name, count, [(prop name, type) | (prop name, count type, item type)]]
        while True:
            line = f.readline()
            if not line:
                raise ValueError("%s: PLY header has no end_header" % path)
            w = line.decode("ascii", "replace").split()
            if not w or w[0] in ("comment", "obj_info"):
                continue
            if w[0] == "format":
                fmt = w[1]
            elif w[0] == "element":
                elems.append([w[1], int(w[2]), []])
            elif w[0] == "property":
                elems[-1][2].append((w[-1], w[2], w[3]) if w[1] == "list" else (w[-1], w[1]))
            elif w[0] == "end_header":
                break
        if fmt not in ("ascii", "binary_little_endian", "binary_big_endian"):
            raise ValueError("%s: PLY format %r" % (path, fmt))
        body = f.read()
    verts = tris = colors = None
    end = "<" if fmt != "binary_big_endian" else ">"
    if fmt == "ascii":
        tok = body.split()
        pos = 0
    else:
        pos = 0
    for name, cnt, props in elems:
        scalar = all(len(p) == 2 for p in props)
        if fmt == "ascii":
            if scalar:
                arr = np.array(tok[pos : pos + cnt * len(props)], dtype=np.float64).reshape(cnt, len(props))
                pos += cnt * len(props)
                cols = {p[0]: arr[:, i] for i, p in enumerate(props)}
            else:
                if len(props) != 1:
                    raise ValueError("%s: element %s mixes list and scalar properties" % (path, name))
                rows = []
                for _ in range(cnt):
                    k = int(tok[pos])
                    rows.append(tok[pos + 1 : pos + 1 + k])
                    pos += 1 + k
                cols = {props[0][0]: rows}
        else:
            if scalar:
                dt = np.dtype([(p[0], end + _PLY_TYPES[p[1]]) for p in props])
                arr = np.frombuffer(body, dt, cnt, pos)
                pos += cnt * dt.itemsize
                cols = {p[0]: arr[p[0]] for p in props}
            else:
                if len(props) != 1:
                    raise ValueError("%s: element %s mixes list and scalar properties" % (path, name))
                ct, it = np.dtype(end + _PLY_TYPES[props[0][1]]), np.dtype(end + _PLY_TYPES[props[0][2]])
                k0 = int(np.frombuffer(body, ct, 1, pos)[0]) if cnt else 0
                rec = np.dtype([("n", ct), ("v", it, (k0,))])
                arr = np.frombuffer(body, rec, cnt, pos)  # the common case: every list has the first one's length (checked below)
                if cnt and not (arr["n"] == k0).all():
                    rows, q = [], pos
                    for _ in range(cnt):
                        k = int(np.frombuffer(body, ct, 1, q)[0])
                        rows.append(np.frombuffer(body, it, k, q + ct.itemsize))
                        q += ct.itemsize + k * it.itemsize
                    pos, cols = q, {props[0][0]: rows}
                else:
                    pos += cnt * rec.itemsize
                    cols = {props[0][0]: arr["v"]}
        if name == "vertex":
            verts = np.stack([np.asarray(cols[a], dtype=np.float64) for a in ("x", "y", "z")], axis=1).astype(F)
            if len(props) > 5 and all(c in cols for c in ("red", "green", "blue")):
                colors = (np.stack([np.asarray(cols[c], dtype=np.float64) for c in ("red", "green", "blue")], axis=1) / 255.0).astype(F)
        elif name == "face":
            rows = cols[props[0][0]]
            if any(len(r) != 3 for r in rows):
                raise ValueError("%s: faces must be triangles (PlyReader.cpp:163-167 reads three indices)" % path)
            tris = np.asarray(rows, dtype=np.int64).reshape(cnt, 3).astype(np.int32)
    if verts is None or tris is None:
        raise ValueError("%s: PLY needs a vertex and a face element" % path)
    if tris.size and (tris.min() < 0 or tris.max() >= len(verts)):
        raise ValueError("%s: face index out of range" % path)
    return verts, tris, colors


def read_ply(path):
    """(verts, tris) of a PLY file (data/geom/bunny/reconstruction/*.ply); colours: read_ply_full."""
    v, t, _ = read_ply_full(path)
    return v, t


def load_mesh_file(path):
    if path.endswith(".npz"):
        d = np.load(path)
        return d["verts"].astype(F), d["tris"].astype(np.int32)
    if path.endswith(".ply"):
        return read_ply(path)
    return read_obj(path)


GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


# ------------------------------------------------------------------ golden scene 2: bunny.obj
def bunny_scene(width=512, height=512, path=None):
    """src/apps/render/SimpleFileLoadApp.cpp:186-241 (golden bunny.ppm): identity instance, default material."""
    v, t = load_mesh_file(path or os.path.join(GOLDEN_DIR, "bunny.obj"))
    mesh = MeshData(v, t, default_material())
    cam = Camera((0.0, 0.1, 0.3), (0.0, 0.1, -0.3), (0.0, 1.0, 0.0), float(F(45.0 * np.pi / 180.0)), width, height, 1, 1, 0.0)
    return _assemble([mesh], [0], [mat_translate_scale((0, 0, 0), (1, 1, 1))], point_light((0.0, 0.1, 0.5)), cam, "bunny")


def bunny70k_scene(width=1920, height=1080, path=None):
    """BASELINE config 2: bun_zipper (69,451 tris), single domain, depth 1, camera framing the bbox."""
    v, t = load_mesh_file(path or os.path.join(GOLDEN_DIR, "bun_zipper.npz"))
    mesh = MeshData(v, t, default_material())
    lo, hi = mesh.bbox()
    c = 0.5 * (lo + hi)
    cam = Camera((float(c[0]), float(c[1]), float(c[2] + 0.35)), (float(c[0]), float(c[1]), float(c[2])), (0.0, 1.0, 0.0),
                 float(F(30.0 * np.pi / 180.0)), width, height, 1, 1, 0.0)
    light = point_light((float(c[0]), float(c[1] + 0.1), float(c[2] + 0.5)))
    return _assemble([mesh], [0], [mat_translate_scale((0, 0, 0), (1, 1, 1))], light, cam, "bunny70k")


def bunny_grid_scene(nx=4, ny=2, pitch=0.3, width=1900, height=1080, path=None):
    """BASELINE config 4: nx*ny bunny instances on a grid (pitch like data/bunny.conf:27-29, film data/bunny.conf:8)."""
    v, t = load_mesh_file(path or os.path.join(GOLDEN_DIR, "bunny.obj"))
    mesh = MeshData(v, t, default_material())
    mats = []
    for j in range(ny):
        for i in range(nx):
            mats.append(mat_translate_scale(((i - (nx - 1) / 2) * pitch, (j - (ny - 1) / 2) * pitch, 0.0), (1, 1, 1)))
    cam = Camera((0.0, 0.1, 1.6), (0.0, 0.1, 0.0), (0.0, 1.0, 0.0), float(F(40.0 * np.pi / 180.0)), width, height, 1, 1, 0.0)
    return _assemble([mesh], [0] * (nx * ny), mats, point_light((0.9, 0.25, 0.25)), cam, "bunny-grid-%dx%d" % (nx, ny))  # light low and to the side: shadow rays cross neighbouring domains


# ------------------------------------------------------------------ synthetic soups
def triangle_soup(n_tris, seed=12345, half_extent=0.005):
    """BASELINE config 3 geometry: centres U[0,1]^3, 3 offsets U[-s,s]^3 each, unshared vertices.
    Generator: numpy Philox(seed), float32 draws."""
    rng = np.random.Generator(np.random.Philox(seed))
    c = rng.random((n_tris, 1, 3), dtype=F)
    o = rng.random((n_tris, 3, 3), dtype=F)  # (in place from here: the same float32 operations in the same order, no 360 MB temporaries)
    o *= F(2.0); o -= F(1.0); o *= F(half_extent); o += c
    tris = np.arange(n_tris * 3, dtype=np.int32).reshape(-1, 3)
    return o.reshape(-1, 3), tris


def soup_scene(n_tris=10_000_000, width=1920, height=1080, seed=12345, half_extent=0.005):
    """BASELINE config 3: eye (.5,.5,3) -> (.5,.5,.5), fov 30 deg, light at the eye, kd .5, depth 1."""
    v, t = triangle_soup(n_tris, seed, half_extent)
    mesh = MeshData(v, t, default_material())
    cam = Camera((0.5, 0.5, 3.0), (0.5, 0.5, 0.5), (0.0, 1.0, 0.0), float(F(30.0 * np.pi / 180.0)), width, height, 1, 1, 0.0)
    return _assemble([mesh], [0], [mat_translate_scale((0, 0, 0), (1, 1, 1))], point_light((0.5, 0.5, 3.0)), cam,
                     "soup-%d" % n_tris)


def cathedral_scene(width=512, height=512, samples=2, depth=2, seed=7, eye=(0.0, 1.5, 9.0), light=(0.0, 2.5, 4.0)):
    """BASELINE config 5 stand-in (sibenik.obj is missing from the reference, .MISSING_LARGE_BLOBS):
    a hall of long thin triangles (~80 K; floor, ceiling, side walls, back wall, two rows of columns; open at z = +10) --
    deep BVH, AO-style secondary rays.  The default eye stands inside the hall (adapter-level tests: the schedulers never
    assign a ray to an instance box it starts in, RayPacket.h:195-197); whole-frame tests look in from outside, e.g.
    eye=(0, 1.5, 13), with the light in front of the opening, e.g. light=(0, 2.5, 12): a shadow ray is occluded by anything on the
    whole half-line (tfar = FLT_MAX, EmbreeMeshAdapter.cpp:364-385), so a light inside the hall lights next to nothing."""
    rng = np.random.Generator(np.random.Philox(seed))
    vs, fs = [], []

    def quad_strip(p0, du, dv, nu, nv):
        base = len(vs)
        for j in range(nv + 1):
            for i in range(nu + 1):
                vs.append(p0 + du * (i / nu) + dv * (j / nv))
        for j in range(nv):
            for i in range(nu):
                a = base + j * (nu + 1) + i
                fs.append((a, a + 1, a + nu + 2))
                fs.append((a, a + nu + 2, a + nu + 1))

    X, Y, Z = 4.0, 3.0, 10.0
    e = lambda *a: np.array(a, np.float64)
    n_long, n_thin = 4, 1250  # long thin slivers along z
    quad_strip(e(-X, 0, -Z), e(2 * X, 0, 0), e(0, 0, 2 * Z), n_thin, n_long)  # floor
    quad_strip(e(-X, Y, -Z), e(0, 0, 2 * Z), e(2 * X, 0, 0), n_long, n_thin)  # ceiling
    quad_strip(e(-X, 0, -Z), e(0, 0, 2 * Z), e(0, Y, 0), n_long, n_thin)  # left wall
    quad_strip(e(X, 0, -Z), e(0, Y, 0), e(0, 0, 2 * Z), n_thin, n_long)  # right wall
    quad_strip(e(-X, 0, -Z), e(0, Y, 0), e(2 * X, 0, 0), 50, 50)  # back
    for k in range(12):  # columns: thin tall quads
        cx = (-1 if k % 2 else 1) * 2.0
        cz = -8.0 + (k // 2) * 3.0
        for s in range(16):
            a0, a1 = 2 * np.pi * s / 16, 2 * np.pi * (s + 1) / 16
            p0 = e(cx + 0.3 * np.cos(a0), 0, cz + 0.3 * np.sin(a0))
            p1 = e(cx + 0.3 * np.cos(a1), 0, cz + 0.3 * np.sin(a1))
            quad_strip(p0, p1 - p0, e(0, Y, 0), 1, 150)
    v = np.array(vs, F)
    v += (rng.random(v.shape, dtype=F) - F(0.5)) * F(1e-3)
    mesh = MeshData(v, np.array(fs, np.int32), default_material(kd=(0.8, 0.8, 0.8)))
    cam = Camera(tuple(eye), (0.0, 1.4, 0.0), (0.0, 1.0, 0.0), float(F(60.0 * np.pi / 180.0)), width, height, samples, depth, 0.0)
    return _assemble([mesh], [0], [mat_translate_scale((0, 0, 0), (1, 1, 1))], point_light(tuple(light)), cam, "cathedral")


def split_into_domains(scene, n_domains, axis=2):
    """A one-mesh scene cut into n_domains spatial domains along `axis` (triangles assigned by centroid slab, vertices
    re-indexed per domain; one Mesh + the original instance matrix each) -- what a GraviT application does when it hands the
    Domain scheduler one mesh per rank (DomainTracer.h:115-144)."""
    assert len(scene.meshes) == 1 and scene.n_inst == 1
    m = scene.meshes[0]
    lo, hi = m.bbox()
    c = m.verts[m.tris].mean(axis=1)[:, axis]
    cell = np.clip(((c - lo[axis]) / max(float(hi[axis] - lo[axis]), 1e-30) * n_domains).astype(np.int64), 0, n_domains - 1)
    meshes, mats = [], []
    for d in range(n_domains):
        sel = m.tris[cell == d]
        used, inv = np.unique(sel.reshape(-1), return_inverse=True)
        meshes.append(MeshData(np.ascontiguousarray(m.verts[used]), inv.astype(np.int32).reshape(-1, 3), m.material))
        mats.append(scene.m[0].reshape(16))
    return _assemble(meshes, list(range(n_domains)), mats, scene.lights, scene.camera, "%s-dom%d" % (scene.name, n_domains))


# ------------------------------------------------------------------ domain decomposition of the soup
def domain_grid(n_domains):
    """x-y tiling first (rays of the config-3 camera travel along -z), then z."""
    return {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (4, 2, 1), 16: (4, 4, 1)}.get(n_domains, (n_domains, 1, 1))


def soup_domains_scene(n_tris=10_000_000, n_domains=8, width=1920, height=1080, seed=12345, half_extent=0.005):
    """The config-3 soup cut into n_domains spatial domains (one Mesh + identity instance each, triangles assigned
    by centroid cell), the layout GraviT's Domain scheduler distributes over ranks (DomainTracer.h:115-144)."""
    v, t = triangle_soup(n_tris, seed, half_extent)
    gx, gy, gz = domain_grid(n_domains)
    tv = v.reshape(-1, 3, 3)

    def cell_of(k, g):  # the centroid's cell along axis k: float32 mean over the three vertices, as np.mean(axis=1) adds them (v0 + v1 + v2, then / 3)
        if g == 1:
            return 0
        ck = (tv[:, 0, k] + tv[:, 1, k] + tv[:, 2, k]) / F(3.0)
        return np.clip((ck * g).astype(np.int32), 0, g - 1)

    cell = (cell_of(2, gz) * gy + cell_of(1, gy)) * gx + cell_of(0, gx)
    meshes, mats = [], []
    tri_verts = v.reshape(-1, 3, 3)
    for d in range(gx * gy * gz):
        sel = np.nonzero(cell == d)[0]
        dv = np.ascontiguousarray(tri_verts[sel].reshape(-1, 3))
        dt = np.arange(len(sel) * 3, dtype=np.int32).reshape(-1, 3)
        meshes.append(MeshData(dv, dt, default_material()))
        mats.append(mat_translate_scale((0, 0, 0), (1, 1, 1)))
    cam = Camera((0.5, 0.5, 3.0), (0.5, 0.5, 0.5), (0.0, 1.0, 0.0), float(F(30.0 * np.pi / 180.0)), width, height, 1, 1, 0.0)
    return _assemble(meshes, list(range(len(meshes))), mats, point_light((0.5, 0.5, 3.0)), cam, "soup-%d-dom%d" % (n_tris, n_domains))


def soup_weak_tile(d, n_tiles, n_per_tile, seed=12345, half_extent=None):
    """Tile d of the WEAK-scaling soup: n_per_tile triangles of its own (Philox(seed + 7919 (d + 1))) with centres uniform in cell d of
    domain_grid(n_tiles) -- every rank generates only what it owns.  The triangle size shrinks with the density (half_extent =
    0.005 n_tiles^(-1/3) unless given), so that the scene keeps config 3's extent-to-spacing ratio."""
    gx, gy, gz = domain_grid(n_tiles)
    he = F(0.005 * n_tiles ** (-1.0 / 3.0)) if half_extent is None else F(half_extent)
    rng = np.random.Generator(np.random.Philox(seed + 7919 * (d + 1)))
    ix, iy, iz = d % gx, (d // gx) % gy, d // (gx * gy)
    c = rng.random((n_per_tile, 1, 3), dtype=F) * np.array([1.0 / gx, 1.0 / gy, 1.0 / gz], F) + np.array([ix / gx, iy / gy, iz / gz], F)
    o = rng.random((n_per_tile, 3, 3), dtype=F)
    o *= F(2.0); o -= F(1.0); o *= he; o += c
    return o.reshape(-1, 3), np.arange(n_per_tile * 3, dtype=np.int32).reshape(-1, 3)


def soup_weak_scene(n_per_tile, n_tiles, width, height, own=None, boxes=None, seed=12345, half_extent=None):
    """Weak scaling of config 3: n_tiles domains of n_per_tile triangles EACH (camera and light of config 3, film as given).  own: the
    tiles whose geometry this process generates (None: all); boxes[d] = (lo, hi) of every tile it does not (a Domain-scheduler rank
    knows every instance's box but holds only its own meshes, DomainTracer.h:115-144) -- meshes[d] is None there."""
    own = list(range(n_tiles)) if own is None else list(own)
    meshes, lo, hi, mats = [None] * n_tiles, [None] * n_tiles, [None] * n_tiles, []
    for d in range(n_tiles):
        if d in own:
            v, t = soup_weak_tile(d, n_tiles, n_per_tile, seed, half_extent)
            meshes[d] = MeshData(v, t, default_material())
            lo[d], hi[d] = meshes[d].bbox()
        else:
            lo[d], hi[d] = np.asarray(boxes[d][0], F), np.asarray(boxes[d][1], F)
        mats.append(mat_translate_scale((0, 0, 0), (1, 1, 1)))
    minv, normi = zip(*[instance_matrices(m) for m in mats])
    cam = Camera((0.5, 0.5, 3.0), (0.5, 0.5, 0.5), (0.0, 1.0, 0.0), float(F(30.0 * np.pi / 180.0)), width, height, 1, 1, 0.0)
    return Scene(meshes, list(range(n_tiles)), np.array(mats, F).reshape(-1, 16), np.array(minv, F).reshape(-1, 16), np.array(normi, F).reshape(-1, 9),
                 np.array(lo, F).reshape(-1, 3), np.array(hi, F).reshape(-1, 3), np.ascontiguousarray(point_light((0.5, 0.5, 3.0)), LIGHT_DTYPE), cam,
                 "soup-weak-%dx%d" % (n_tiles, n_per_tile))


# ------------------------------------------------------------------ legacy .conf scenes (data/bunny.conf, data/README.conf)
def load_conf(path, geom_dirs=None, width=None, height=None):
    """Scene from one of the reference's legacy .conf files, as src/apps/render/ConfigFileLoader.cpp:70-285 reads them
    (that loader is compiled out in the reference, `#if 0` at :31, so its quirks are restated from the source):
      F w h                      film size                                         (:105-109)
      C eye(3) look(3) up(3) fov camera; the fov field is parsed but 45 degrees is stored ("TODO", :111-129)
      G file t(3) r(3) s(3)      one instance per line: translate, then scale; the rotation is computed into a temporary
                                 and never applied (:171-186); a mesh file is loaded once and shared (:139-166)
      LP pos(3) color(3)         point light (:225-238);  LA pos(3) normal(3) w h color(3)  area light (:240-262)
      ST DOMAIN | ...            scheduler hint, returned as scene.name suffix
    Mesh paths are resolved by base name in geom_dirs (default: tests/golden)."""
    geom_dirs = list(geom_dirs or [GOLDEN_DIR])
    meshes, mesh_index, inst_mesh, mats, lights = [], {}, [], [], []
    cam = None
    film = [512, 512]
    sched = "image"
    with open(path) as f:
        for line in f:
            if line.startswith("#"):
                continue
            e = line.split()
            if not e:
                continue
            if e[0] == "F" and len(e) >= 3:
                film = [int(e[1]), int(e[2])]
            elif e[0] == "C" and len(e) >= 11:
                v = [float(x) for x in e[1:10]]
                cam = (tuple(v[0:3]), tuple(v[3:6]), tuple(v[6:9]))
            elif e[0] == "G" and len(e) >= 11:
                base = os.path.basename(e[1])
                if base not in mesh_index:
                    found = None
                    for d in geom_dirs:
                        for cand in (os.path.join(d, base), os.path.join(d, os.path.splitext(base)[0] + ".npz")):
                            if os.path.exists(cand):
                                found = cand
                                break
                        if found:
                            break
                    if not found:
                        raise FileNotFoundError("%s: mesh %s not found in %s" % (path, e[1], geom_dirs))
                    v, t = load_mesh_file(found)
                    mesh_index[base] = len(meshes)
                    meshes.append(MeshData(v, t, default_material()))
                tr = [float(x) for x in e[2:5]]
                sc = [float(x) for x in e[8:11]]
                if not any(sc):
                    sc = [1.0, 1.0, 1.0]  # glm::length(t) > 0 guard (:188-192)
                mats.append(mat_translate_scale(tr, sc))
                inst_mesh.append(mesh_index[base])
            elif e[0] == "LP" and len(e) >= 7:
                lights.append(point_light([float(x) for x in e[1:4]], [float(x) for x in e[4:7]]))
            elif e[0] == "LA" and len(e) >= 12:
                from .layouts import area_light

                lights.append(area_light([float(x) for x in e[1:4]], [float(x) for x in e[9:12]], [float(x) for x in e[4:7]],
                                         float(e[7]), float(e[8])))
            elif e[0] == "ST" and len(e) >= 2 and e[1] in ("DOMAIN", "HYBRID"):
                sched = e[1].lower()
    if cam is None or not meshes:
        raise ValueError("%s: no camera or no geometry" % path)
    w, h = (width or film[0]), (height or film[1])
    camera = Camera(cam[0], cam[1], cam[2], float(F(45.0 * np.pi / 180.0)), w, h, 1, 1, 0.0)
    lights = np.concatenate(lights) if lights else np.zeros(0, LIGHT_DTYPE)
    return _assemble(meshes, inst_mesh, mats, np.ascontiguousarray(lights, LIGHT_DTYPE), camera,
                     "%s[%s]" % (os.path.basename(path), sched))
