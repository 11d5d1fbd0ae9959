"""CPU-side checks of the host logic and of the C-ABI library (no device calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from gravit_amd import capi, layouts, scenes
from oracle import orc
from tests.conftest import GOLDEN, ROOT
from tests.helpers import oracle_meshes, rays_equal_bits, seeded_rays_at


def test_layouts_are_the_reference_layouts():
    assert layouts.RAY_DTYPE.itemsize == 80 and layouts.MATERIAL_DTYPE.itemsize == 92 and layouts.LIGHT_DTYPE.itemsize == 64
    assert layouts.RAY_DTYPE.fields["direction"][1] == 16 and layouts.RAY_DTYPE.fields["type"][1] == 60
    assert layouts.MATERIAL_DTYPE.fields["ks"][1] == 16 and layouts.MATERIAL_DTYPE.fields["kd"][1] == 28  # note the order
    assert layouts.RAY_DTYPE == orc.RAY_DTYPE and layouts.MATERIAL_DTYPE == orc.MATERIAL_DTYPE and layouts.LIGHT_DTYPE == orc.LIGHT_DTYPE
    assert layouts.default_material().tobytes() == orc.default_material().tobytes()


def test_c_abi_library_exports_every_declared_symbol():
    """The library loads without a GPU and exports exactly what include/gvt_hip.h declares."""
    hdr = open(os.path.join(ROOT, "include", "gvt_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(gvt_hip_[a-z_0-9]+)\s*\(", hdr)))
    assert len(declared) >= 30
    assert sorted(capi.SYMBOLS) == declared
    lib = capi.load()
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.gvt_hip_abi_version() == capi.ABI_VERSION == int(re.search(r"#define GVT_HIP_ABI_VERSION (\d+)", hdr).group(1))
    # struct sizes the ABI promises
    assert C.sizeof(capi.MeshInfo) == 96 and C.sizeof(capi.Stats) == 16 * 8
    m = re.search(r"typedef struct gvt_hip_ray \{(.*?)\} gvt_hip_ray;", hdr, re.S)
    assert m and "float pad[4]" in m.group(1)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(capi.GvtHipError):
        capi.load()


def test_no_device_is_an_error_not_a_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    lib = capi.load()
    rc = lib.gvt_hip_init(0)
    assert rc != 0 and b"HIP" in lib.gvt_hip_last_error() or b"device" in lib.gvt_hip_last_error()
    assert not lib.gvt_hip_queue_create(C.c_size_t(16))  # every entry point refuses without a device


def test_product_never_imports_the_oracle():
    pat = re.compile(r"^\s*(import\s+oracle|from\s+oracle|from\s+\.+oracle|#\s*include\s*[<\"].*oracle)|liboracle|libgvtref|libsimd_baseline|orc\.py", re.M)
    for root, _, files in os.walk(os.path.join(ROOT, "gravit_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f), errors="replace").read()
                assert not pat.search(src), f


def test_legacy_conf_loader():
    """data/bunny.conf (copied to tests/golden as an input fixture): 3 bunny instances, one shared mesh, 1900x1080 film."""
    sc = scenes.load_conf(os.path.join(GOLDEN, "bunny.conf"))
    assert sc.n_inst == 3 and len(sc.meshes) == 1 and sc.inst_mesh == [0, 0, 0]
    assert (sc.camera.width, sc.camera.height) == (1900, 1080) and sc.name.endswith("[domain]")
    assert np.allclose(sc.m[0].reshape(4, 4)[3, :3], (-0.3, 0, 0)) and np.allclose(sc.m[2].reshape(4, 4)[3, :3], (0.3, 0, 0))
    assert len(sc.lights) == 1 and np.allclose(sc.lights["position"][0], (0.0, 0.1, 0.25))
    assert np.isclose(sc.camera.fov, 45.0 * np.pi / 180.0)  # the loader stores 45 degrees whatever the file says


def test_scenes():
    s = scenes.simple_scene()
    assert s.n_inst == 25 and s.inst_mesh[:4] == [0, 1, 0, 1]
    assert np.allclose(s.m[0].reshape(4, 4)[3], (0, -1.0, -1.0, 1)) and np.isclose(s.m[0][0], 0.4)
    assert np.allclose(s.minv[0][0], 2.5) and np.allclose(s.normi[0][0], 2.5)
    b = scenes.bunny_scene()
    assert b.meshes[0].tris.shape == (4968, 3)
    z = scenes.bunny70k_scene(64, 36)
    assert z.meshes[0].tris.shape == (69451, 3) and z.meshes[0].verts.shape == (35947, 3)
    v, t = scenes.triangle_soup(1000)
    v2, _ = scenes.triangle_soup(1000)
    assert np.array_equal(v, v2) and v.shape == (3000, 3) and v.dtype == np.float32
    assert np.abs(v.reshape(-1, 3, 3) - v.reshape(-1, 3, 3).mean(1, keepdims=True)).max() <= 0.01
    d = scenes.soup_domains_scene(4000, 4, 64, 36)
    assert d.n_inst == 4 and sum(m.tris.shape[0] for m in d.meshes) == 4000
    g = scenes.bunny_grid_scene()
    assert g.n_inst == 8
    c = scenes.cathedral_scene(32, 32)
    assert 70_000 < c.meshes[0].tris.shape[0] < 120_000


@pytest.mark.parametrize("builder", [scenes.bunny_scene, lambda: scenes.soup_scene(20000, 64, 36)])
def test_oracle_bvh_equals_brute_force(builder):
    """The oracle's BVH is conservative: closest/any hits equal the brute-force scan over every triangle, bit for bit."""
    sc = builder()
    om = oracle_meshes(sc)[0]
    lo, hi = om.bbox()
    org, d = seeded_rays_at(lo, hi, 1500, 11)
    a, b = om.intersect(org, d, use_bvh=True), om.intersect(org, d, use_bvh=False)
    assert (a["prim"] == b["prim"]).all() and (a["t"].view(np.uint32) == b["t"].view(np.uint32)).all()
    assert (a["u"].view(np.uint32) == b["u"].view(np.uint32)).all() and (a["v"].view(np.uint32) == b["v"].view(np.uint32)).all()
    assert (a["prim"] >= 0).sum() > 100
    assert (om.occluded(org, d, use_bvh=True) == om.occluded(org, d, use_bvh=False)).all()


def test_oracle_axis_aligned_geometry_edges():
    """Flat, axis-aligned triangles (zero-thickness boxes) and rays through shared edges/vertices."""
    sc = scenes.simple_scene()
    cube = oracle_meshes(sc)[1]
    g = np.linspace(-0.5, 0.5, 21, dtype=np.float32)
    xx, yy = np.meshgrid(g, g)
    org = np.stack([xx.ravel(), yy.ravel(), np.full(xx.size, 2.0, np.float32)], 1)
    d = np.tile(np.array([0, 0, -1], np.float32), (len(org), 1))
    a, b = cube.intersect(org, d, use_bvh=True), cube.intersect(org, d, use_bvh=False)
    assert (a["prim"] == b["prim"]).all() and (a["t"] == b["t"]).all()
    assert (a["prim"] >= 0).all() and np.allclose(a["t"], 1.5)


def test_oracle_trace_output_contract():
    """moved_rays = misses (unchanged) + un-occluded shadow rays (type 1, t_max 3.0, id/w/depth/t copied); rayList.t updated in place."""
    sc = scenes.bunny_scene(64, 64)
    om = oracle_meshes(sc)[0]
    c = sc.camera
    rays = orc.camera_rays(c.eye, c.focus, c.up, c.fov, c.width, c.height)
    before = rays.copy()
    out = om.trace(rays, sc.m[0], sc.minv[0], sc.normi[0], sc.lights, 0)
    miss = out[out["type"] == 0]
    shadow = out[out["type"] == 1]
    assert len(miss) + len(shadow) == len(out) and len(shadow) > 100
    assert (shadow["t_max"] == np.float32(3.0)).all() and (shadow["depth"] == 1).all() and (shadow["w"] == 1.0).all()
    hit_mask = rays["t"] != before["t"]
    assert hit_mask.sum() >= len(shadow) and len(miss) == (~hit_mask).sum()
    assert set(shadow["id"]).issubset(set(before["id"][hit_mask]))
    # threads do not change the result
    rays2 = before.copy()
    out2 = om.trace(rays2, sc.m[0], sc.minv[0], sc.normi[0], sc.lights, 0, nthreads=4)
    assert rays_equal_bits(out, out2)  # (padding bytes 64..79 of the 80-byte record are not compared)
    # empty and ragged ranges
    assert len(om.trace(before.copy()[:0], sc.m[0], sc.minv[0], sc.normi[0], sc.lights, 0)) == 0
    r3 = before.copy()
    out3 = om.trace(r3, sc.m[0], sc.minv[0], sc.normi[0], sc.lights, 0, begin=100, end=1177)
    assert (r3[:100]["t"] == before[:100]["t"]).all() and (r3[1177:]["t"] == before[1177:]["t"]).all() and len(out3) > 0


def test_simd_cpu_baseline_library_loads_and_answers_a_tiny_tree():
    """oracle/libsimd_baseline.so (bench.py's SIMD CPU baseline; measurement infrastructure, not the product, not the oracle) loads, exports its two
    entry points and walks a hand-made one-node tree: one triangle in the unit square at z = 0, rays down the z axis."""
    from oracle import simd

    lib = simd.load()
    assert lib.simd_intersect and lib.simd_occluded
    node = np.zeros(16, np.uint32)
    f = node.view(np.float32)
    f[0], f[1], f[2], f[3] = -1.0, -1.0, -1.0, 2.0 / 255.0   # origin, step.x
    f[14], f[15] = 2.0 / 255.0, 2.0 / 255.0                   # step.y, step.z
    lo = np.array([0, 255, 255, 255], np.uint8).view(np.uint32)[0]      # child 0: the whole grid; children 1..3 inverted (unused)
    hi = np.array([255, 0, 0, 0], np.uint8).view(np.uint32)[0]
    node[4], node[5], node[6], node[7], node[8], node[9] = lo, hi, lo, hi, lo, hi
    node[10] = np.uint32((~((0 << 3) | 1)) & 0xFFFFFFFF)                   # leaf: slot 0, one triangle
    node[11] = node[12] = node[13] = np.uint32(0xFFFFFFFF)               # empty refs
    v0, v1, v2 = np.array([0, 0, 0], np.float32), np.array([1, 0, 0], np.float32), np.array([0, 1, 0], np.float32)
    slot = np.zeros(16, np.float32)
    slot[0:3] = v0; slot.view(np.int32)[3] = 7; slot[4:7] = v0 - v1; slot[8:11] = v2 - v0
    T = simd.Tree(node.reshape(1, 16), slot.reshape(1, 16))
    org = np.array([[0.25, 0.25, 1.0], [0.9, 0.9, 1.0]], np.float32)
    d = np.array([[0, 0, -1.0], [0, 0, -1.0]], np.float32)
    t, prim, u, v = T.intersect(org, d, nthreads=2)
    assert prim.tolist() == [7, -1] and t[0] == np.float32(1.0) and (u[0], v[0]) == (np.float32(0.25), np.float32(0.25))
    assert T.occluded(org, d).tolist() == [True, False]


def test_cost_optimal_wide_collapse_tool_on_a_random_tree(tmp_path):
    """tools/wide_dp.c (the measurement tool behind profiles/r05_wide_dp.txt): on a random binary tree the dynamic programme's collapse never has
    more than W children per wide node, covers the tree (every leaf is reached through marked roots), costs no more than the greedy collapse the
    4-wide layout is built with, and its PLOC rebuild returns a tree over the same leaves."""
    import ctypes as C
    import subprocess
    import sys

    so = str(tmp_path / "libwide_dp.so")
    subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "wide_dp.c"), "-lm"])
    lib = C.CDLL(so)
    rng = np.random.default_rng(7)
    n_leaves = 2000
    pts = rng.random((n_leaves, 3)).astype(np.float32)
    nodes = []
    sys.setrecursionlimit(20000)

    def rec(idx):  # random-split binary tree; returns (ref, lo, hi); leaf refs are negative like the library's
        if len(idx) == 1:
            return ~int(idx[0] * 8 + 1), pts[idx[0]] - 0.01, pts[idx[0]] + 0.01
        ax = int(np.argmax(pts[idx].max(0) - pts[idx].min(0)))
        order = idx[np.argsort(pts[idx, ax])]
        h = int(rng.integers(1, len(order)))
        me = len(nodes)
        nodes.append(None)
        a, b = rec(order[:h]), rec(order[h:])
        row = np.zeros(16, np.float32)
        for s_, (_, lo, hi) in enumerate((a, b)):
            row[4 * s_:4 * s_ + 4] = (lo[0], hi[0], lo[1], hi[1])
            row[8 + 2 * s_:10 + 2 * s_] = (lo[2], hi[2])
        row.view(np.int32)[12:14] = (a[0], b[0])
        nodes[me] = row
        return me, np.minimum(a[1], b[1]), np.maximum(a[2], b[2])

    rec(np.arange(n_leaves))
    tree = np.ascontiguousarray(np.stack(nodes))
    p = lambda a: a.ctypes.data_as(C.c_void_p)  # noqa: E731

    def sah(marks):  # sum of the marked nodes' areas (the root's from its children's union), as the tool defines its cost
        area = np.zeros(len(tree))
        r = tree[0]
        d = [max(r[1], r[5]) - min(r[0], r[4]), max(r[3], r[7]) - min(r[2], r[6]), max(r[9], r[11]) - min(r[8], r[10])]
        area[0] = d[0] * d[1] + d[1] * d[2] + d[2] * d[0]
        for k, row in enumerate(tree):
            c = row.view(np.int32)[12:14]
            for s_ in range(2):
                if c[s_] >= 0:
                    dx, dy, dz = row[4 * s_ + 1] - row[4 * s_], row[4 * s_ + 3] - row[4 * s_ + 2], row[9 + 2 * s_] - row[8 + 2 * s_]
                    area[c[s_]] = dx * dy + dy * dz + dz * dx
        return float(area[marks.astype(bool)].sum() / area[0])

    for W in (4, 8):
        marks, greedy, st = np.zeros(len(tree), np.uint8), np.zeros(len(tree), np.uint8), np.zeros(4 + 17)
        assert lib.wide_dp(p(tree), C.c_int64(len(tree)), C.c_int(W), p(marks), p(st)) == 0
        assert lib.wide_greedy(p(tree), C.c_int64(len(tree)), C.c_int(W), p(greedy)) == 0
        assert marks[0] == 1 and greedy[0] == 1
        assert st[4 + W + 1:].sum() == 0 and st[4:4 + W + 1].sum() == st[0] == marks.sum()     # every wide node has 2..W children
        assert abs(st[1] - (marks.sum() - 1 + n_leaves)) < 0.5                                  # children = the other roots + all leaves: the tree is covered
        assert abs(sah(marks) - st[2]) < 1e-3 * st[2] and sah(marks) <= sah(greedy) * (1 + 1e-6)  # never dearer than the greedy collapse
    out = np.zeros_like(tree)
    assert lib.ploc_rebuild(p(tree), C.c_int64(len(tree)), C.c_int(8), p(out)) == 0
    refs = out.view(np.int32)[:, 12:14].reshape(-1)
    assert sorted(refs[refs < 0].tolist()) == sorted(tree.view(np.int32)[:, 12:14].reshape(-1)[tree.view(np.int32)[:, 12:14].reshape(-1) < 0].tolist())
    assert sorted(refs[refs >= 0].tolist()) == list(range(1, len(tree)))                         # every inner node but the root is some node's child


def test_rccl_stand_in_of_the_multi_process_tests_builds_and_covers_the_librarys_imports(tmp_path):
    """tests/fake_rccl (the transport under tests/test_gpu_multiproc.py) compiles against <rccl/rccl.h> and exports every entry point the library
    resolves from librccl (gravit_amd/csrc/domain.hip, GVT_SYM + ncclCommAbort) -- a new import in the library must reach the stand-in too."""
    import re
    import subprocess

    from tests.conftest import ROOT
    so = str(tmp_path / "libfakerccl.so")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-fPIC", "-shared", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", so,
                    os.path.join(ROOT, "tests", "fake_rccl", "fake_rccl.cpp"), "-L/opt/rocm/lib", "-lamdhip64"], check=True, timeout=300)
    exported = set(re.findall(r" T (nccl\w+)", subprocess.run(["nm", "-D", so], check=True, stdout=subprocess.PIPE, text=True).stdout))
    src = open(os.path.join(ROOT, "gravit_amd", "csrc", "domain.hip")).read()
    wanted = set(re.findall(r'"(nccl[A-Za-z]+)"', src))
    assert len(wanted) >= 11 and wanted <= exported, wanted - exported


def test_read_ply_ascii_binary_and_colors(tmp_path):
    """scenes.read_ply_full against tiny committed PLY files in the three encodings PlyReader.cpp:54-176 takes through ply.c; colours only when the
    vertex element has more than five properties (PlyReader.cpp:123), as /255 floats; the reference's bun_zipper.ply (where the GraviT tree is present)
    reproduces the committed bun_zipper.npz."""
    import os

    from gravit_amd import scenes
    from tests.conftest import GOLDEN

    verts = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1.5]], np.float32)
    cols = np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 64, 32], [10, 20, 30]], np.float32) / np.float32(255.0)
    tris = np.array([[0, 1, 2], [0, 2, 3], [1, 4, 2], [3, 2, 4]], np.int32)
    for name in ("tiny_ascii.ply", "tiny_le.ply", "tiny_be.ply"):
        v, t, c = scenes.read_ply_full(os.path.join(GOLDEN, name))
        assert v.dtype == np.float32 and t.dtype == np.int32 and c.dtype == np.float32, name
        assert np.array_equal(v, verts) and np.array_equal(t, tris) and np.array_equal(c, cols), name
    v, t, c = scenes.read_ply_full(os.path.join(GOLDEN, "tiny_nocolor.ply"))
    assert np.array_equal(v, verts) and np.array_equal(t, tris) and c is None
    v2, t2 = scenes.load_mesh_file(os.path.join(GOLDEN, "tiny_le.ply"))
    assert np.array_equal(v2, verts) and np.array_equal(t2, tris)
    bad = tmp_path / "quad.ply"
    bad.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 4\nproperty float x\nproperty float y\nproperty float z\nelement face 1\n"
                    b"property list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n1 1 0\n0 1 0\n4 0 1 2 3\n")
    with pytest.raises(ValueError):
        scenes.read_ply(str(bad))
    oob = tmp_path / "oob.ply"
    oob.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\nelement face 1\n"
                    b"property list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n1 1 0\n3 0 1 7\n")
    with pytest.raises(ValueError):
        scenes.read_ply(str(oob))
    ref = "/root/reference/data/geom/bunny/reconstruction/bun_zipper.ply"
    if os.path.exists(ref):
        z = np.load(os.path.join(GOLDEN, "bun_zipper.npz"))
        v, t = scenes.read_ply(ref)
        keys = list(z.keys())
        vk = [k for k in keys if z[k].dtype.kind == "f" and z[k].ndim == 2][0]
        tk = [k for k in keys if z[k].dtype.kind in "iu" and z[k].ndim == 2][0]
        assert np.array_equal(v, z[vk]) and np.array_equal(t, z[tk].astype(np.int32))
