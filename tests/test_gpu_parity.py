"""Parity tests proper: the HIP adapter (through the C ABI) against the pinned CPU oracle on the same seeded
inputs, against the committed golden fixtures, and -- at the full BASELINE sizes -- through size-independent
properties.  Integer/index results (primID, hit/miss, ray counts, ids) and all Lambert-path floats must be
BIT-EXACT, and so is the secondary-bounce path (its sin / cos / acos are the written-out definitions of include/gvt_math.h on both
sides); Phong/Blinn (powf) and frames with several deposits per pixel (float atomics in arbitrary order) are held to 1e-5 absolute
on radiance, the tolerance BASELINE.json's north_star states.

Run on the GPU box:  python -m pytest tests -m gpu -x -q
"""
import hashlib
import os

import numpy as np
import pytest

from gravit_amd import layouts, scenes
from gravit_amd.adapter import FrameBuffer, HipMeshAdapter, RayQueue, TopLevel, camera_generate
from gravit_amd.layouts import NORMALS_FLAT, NORMALS_SMOOTH, RAY_DTYPE
from gravit_amd.layouts import RAY_EPSILON as RAY_EPSILON_F
from gravit_amd.scheduler import ImageTracer
from oracle import orc
from tests.conftest import GOLDEN, read_ppm
from tests.helpers import bits, oracle_camera_rays, oracle_meshes, oracle_render, rays_equal_bits, seeded_rays_at, sort_rays

pytestmark = pytest.mark.gpu

RADIANCE_TOL = 1e-5  # BASELINE.json north_star: "within 1e-5 on float radiance"


def assert_hits_equal(g, c):
    assert (g["prim"] == c["prim"]).all(), "%d primIDs differ" % (g["prim"] != c["prim"]).sum()
    for f in ("t", "u", "v"):
        assert (bits(g[f]) == bits(c[f])).all(), "%s differs in %d rays" % (f, (bits(g[f]) != bits(c[f])).sum())


# ------------------------------------------------------------------ the two queries under the adapter
def test_golden_hit_vectors_bunny(hip, oracle_vectors):
    """Committed fixture: 4096 seeded rays on bunny.obj, (t, primID, u, v) and occlusion flags from the pinned oracle."""
    sc = scenes.bunny_scene()
    ad = HipMeshAdapter(sc.meshes[0])
    g = ad.intersect(oracle_vectors["org"], oracle_vectors["dirs"])
    assert_hits_equal(g, oracle_vectors["hits"])
    assert (ad.occluded(oracle_vectors["org"], oracle_vectors["dirs"]) == oracle_vectors["occluded"]).all()
    assert (g["prim"] >= 0).sum() > 1000


@pytest.mark.parametrize("name", ["bunny", "bunny70k", "soup200k", "cone", "cube", "cathedral"])
def test_closest_and_any_hit_match_oracle(hip, name):
    mesh = {"bunny": lambda: scenes.bunny_scene().meshes[0], "bunny70k": lambda: scenes.bunny70k_scene().meshes[0],
            "soup200k": lambda: scenes.soup_scene(200_000, 64, 36).meshes[0], "cone": lambda: scenes.simple_scene().meshes[0],
            "cube": lambda: scenes.simple_scene().meshes[1], "cathedral": lambda: scenes.cathedral_scene(32, 32).meshes[0]}[name]()
    ad = HipMeshAdapter(mesh)
    om = orc.Mesh(mesh.verts, mesh.tris)
    lo, hi = om.bbox()
    org, d = seeded_rays_at(lo, hi, 20011, 3)  # ragged: not a multiple of 64
    if name == "cathedral":  # rays from inside the hall
        rng = np.random.default_rng(9)
        org = np.tile(np.array([0.0, 1.5, 5.0], np.float32), (20011, 1))
        d = rng.normal(size=(20011, 3)).astype(np.float32)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
    g, c = ad.intersect(org, d), om.intersect(org, d)
    assert_hits_equal(g, c)
    assert (ad.occluded(org, d) == om.occluded(org, d)).all()
    info = ad.info()
    assert info["n_tris"] == len(mesh.tris) and np.allclose(info["bbox_lo"], lo) and np.allclose(info["bbox_hi"], hi)


@pytest.mark.parametrize("opts", [dict(sort_rays=1), dict(long_steps=2, long_min_rays=0), dict(long_steps=0), dict(term_sink=0),
                                  dict(leaf_max=1), dict(leaf_max=4, long_steps=4, long_min_rays=0), dict(leaf_max=3)])
def test_results_do_not_depend_on_tuning_knobs(hip, opts):
    """Every knob of the shipped library that touches the adapter call -- sorting, parking threshold, sink, list order, leaf size -- returns
    the same bits (the tuned constants -- refill / phase thresholds, grid sizes, drain sharing -- and the variants that lost can be moved
    in the experiments build only: tests/experiment_cases.py sweeps them there)."""
    sc = scenes.soup_scene(150_000, 160, 90)
    mesh = sc.meshes[0]
    om = orc.Mesh(mesh.verts, mesh.tris, mesh_mat=mesh.material)
    lo, hi = om.bbox()
    org, d = seeded_rays_at(lo, hi, 30_001, 21)
    rays = oracle_camera_rays(sc)
    try:
        for k, v in opts.items():
            hip.set_option(k, v)
        ad = HipMeshAdapter(mesh)  # after the options: leaf_max is a build-time knob of the mesh
        assert ad.info()["max_leaf"] == opts.get("leaf_max", 2)
        assert_hits_equal(ad.intersect(org, d), om.intersect(org, d))
        assert (ad.occluded(org, d) == om.occluded(org, d)).all()
        rg, rc = rays.copy(), rays.copy()
        og = ad.trace(rg, sc.m[0], sc.minv[0], sc.normi[0], sc.lights)
        oc = om.trace(rc, sc.m[0], sc.minv[0], sc.normi[0], sc.lights, 0)
        assert rays_equal_bits(sort_rays(og), sort_rays(oc)) and rays_equal_bits(rg, rc)
    finally:
        hip.set_option("defaults", 0)


def test_axis_aligned_faces_edges_and_vertices(hip):
    """Rays through shared edges and vertices of the cube's flat, axis-aligned faces (zero-thickness boxes)."""
    cube = scenes.simple_scene().meshes[1]
    ad, om = HipMeshAdapter(cube), orc.Mesh(cube.verts, cube.tris)
    g = np.linspace(-0.5, 0.5, 33, dtype=np.float32)
    xx, yy = np.meshgrid(g, g)
    for axis in range(3):
        org = np.zeros((xx.size, 3), np.float32)
        org[:, (axis + 1) % 3], org[:, (axis + 2) % 3], org[:, axis] = xx.ravel(), yy.ravel(), 2.0
        d = np.zeros_like(org)
        d[:, axis] = -1.0
        a, b = ad.intersect(org, d), om.intersect(org, d)
        assert_hits_equal(a, b)
        assert (a["prim"] >= 0).all()


def test_degenerate_inputs(hip):
    """Empty ray lists, empty / tiny meshes, zero-area triangles, rays with zero direction components."""
    tri = scenes.MeshData(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2]], np.int32))
    ad = HipMeshAdapter(tri)
    assert len(ad.intersect(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32))) == 0
    h = ad.intersect([[0.25, 0.25, 1.0], [2, 2, 1]], [[0, 0, -1], [0, 0, -1]])
    assert h["prim"].tolist() == [0, -1] and h["t"][0] == 1.0 and np.allclose([h["u"][0], h["v"][0]], 0.25)
    empty = HipMeshAdapter(scenes.MeshData(np.zeros((0, 3), np.float32), np.zeros((0, 3), np.int32)))
    assert empty.intersect([[0, 0, 1]], [[0, 0, -1]])["prim"].tolist() == [-1]
    assert empty.occluded([[0, 0, 1]], [[0, 0, -1]]).tolist() == [0]
    # 5 triangles (one more than a leaf), one of them zero-area, coincident duplicates -> lower primID wins the tie
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 0], [1, 0, 0], [0, 1, 0], [5, 5, 5], [5, 5, 5], [5, 5, 5], [0, 0, -1],
                  [1, 0, -1], [0, 1, -1], [0, 0, -2], [1, 0, -2], [0, 1, -2]], np.float32)
    t = np.arange(15, dtype=np.int32).reshape(5, 3)
    m = scenes.MeshData(v, t)
    a, b = HipMeshAdapter(m), orc.Mesh(v, t)
    org, d = [[0.2, 0.2, 1.0]], [[0, 0, -1]]
    assert a.intersect(org, d)["prim"].tolist() == [0] == b.intersect(org, d)["prim"].tolist()


def test_deep_trees_from_exponentially_spaced_clusters_and_coincident_triangles(hip, capfd, monkeypatch):
    """Sixty clusters of small triangles at 2^-k along x, along y and along z (k = 1 .. 20, each cluster 2^-(k+6) wide) -- the Morton
    order makes a comb, one cluster split off per level -- plus 1,500 coincident copies of one triangle (a subtree split by index alone):
    a tree several times deeper than log4 of its size, so the 4-wide collapse needs more batches of levels than its first (lbvh.hip
    build_nodes4) and the traversal's stack goes a long way down.  Rays start a few dozen cluster widths from their cluster (a ray from
    across the scene cannot resolve a triangle of 1e-9: both sides' results are then rounding noise).  Hits and occlusion flags against
    the oracle, bit for bit; the lower primID wins among the coincident triangles."""
    import re

    rng = np.random.default_rng(5)
    verts, per, sizes = [], 400, []
    for arm in range(3):
        for k in range(1, 21):
            c = np.zeros(3)
            c[arm] = 2.0 ** -k
            sz = 2.0 ** -(k + 6)
            ctr = c + rng.uniform(0, sz, (per, 1, 3))
            verts.append(ctr + rng.uniform(0, sz / 4, (per, 3, 3)))
            sizes.append(sz)
    one = np.array([[0.30, 0.40, 0.45], [0.36, 0.40, 0.45], [0.33, 0.46, 0.45]])
    verts.append(np.broadcast_to(one, (1500, 3, 3)))
    v = np.concatenate(verts).reshape(-1, 3).astype(np.float32)
    t = np.arange(len(v), dtype=np.int32).reshape(-1, 3)
    first_coincident = 60 * per
    monkeypatch.setenv("GVT_HIP_BUILD_TRACE", "1")
    capfd.readouterr()
    ad = HipMeshAdapter(scenes.MeshData(v, t))
    trace = capfd.readouterr().err
    monkeypatch.delenv("GVT_HIP_BUILD_TRACE")
    m = re.search(r"4-wide collapse: (\d+) levels \((\d+) nodes\), (\d+) launched", trace)
    assert m, trace
    levels, nodes4, launched = int(m.group(1)), int(m.group(2)), int(m.group(3))
    n, first_batch = ad.info()["n_nodes"], 3  # the first batch of levels build_nodes4 launches: log4 of the node count + 3
    while n > 1:
        n >>= 2
        first_batch += 1
    assert levels >= 20 and launched > levels >= first_batch, (levels, launched, first_batch)  # more than one batch of levels
    om = orc.Mesh(v, t)
    tri = rng.integers(0, first_coincident, 20_000)
    szr = np.array(sizes)[tri // per][:, None]
    tgt = v.reshape(-1, 3, 3)[tri].mean(axis=1) + rng.uniform(-0.1, 0.1, (20_000, 3)) * szr
    dirs = rng.normal(size=(20_000, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    org = np.concatenate([tgt - dirs * szr * rng.uniform(30, 60, (20_000, 1)), rng.uniform([0.2, 0.3, 1.5], [0.45, 0.55, 2.0], (4_000, 3))]).astype(np.float32)
    tgt = np.concatenate([tgt, np.tile(one.mean(axis=0), (4_000, 1)) + rng.uniform(-0.02, 0.02, (4_000, 3))]).astype(np.float32)
    d = tgt - org
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    a, b = ad.intersect(org, d), om.intersect(org, d)
    assert_hits_equal(a, b)
    assert (ad.occluded(org, d) == om.occluded(org, d)).all()
    small = (a["prim"][:20_000] >= 0) & (tri // per % 20 >= 14)  # hits in the clusters of 2^-15 and below
    assert small.sum() > 2_000 and (a["prim"][:20_000] >= 0).sum() > 12_000
    on_stack = a["prim"][20_000:]
    assert (on_stack[on_stack >= first_coincident] == first_coincident).all() and (on_stack == first_coincident).sum() > 1_000


@pytest.mark.parametrize("seed", range(24))
def test_random_meshes_and_rays_against_the_oracle(hip, seed):
    """Seeded fuzz of the two queries: meshes of 1 .. 3,000 triangles with shared vertices, vertex coordinates on mixed scales, zero-area
    and duplicated triangles, slivers; rays from outside and inside the box, axis-parallel directions (zero components), directions along
    triangle planes and through vertices, ragged counts.  Hits (t, primID, u, v) and occlusion flags against the oracle, bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    n_v = int(rng.integers(3, 2000))
    n_t = int(rng.integers(1, 3000))
    scale = 10.0 ** rng.integers(-3, 3)
    v = rng.normal(size=(n_v, 3)) * scale
    if seed % 3 == 1:  # a few vertices far out / very close together
        v[rng.integers(0, n_v, max(1, n_v // 50))] *= 50.0
        v[rng.integers(0, n_v, max(1, n_v // 50))] = v[0] + rng.normal(size=(max(1, n_v // 50), 3)) * scale * 1e-4
    if seed % 4 == 2:  # axis-aligned structure: many coplanar triangles, zero-thickness boxes
        v[:, int(rng.integers(0, 3))] = np.round(v[:, 0] / scale) * scale
    v = v.astype(np.float32)
    t = rng.integers(0, n_v, (n_t, 3)).astype(np.int32)  # index triples may repeat a vertex: zero-area triangles
    if n_t > 4:
        t[rng.integers(0, n_t, n_t // 8 + 1)] = t[rng.integers(0, n_t, n_t // 8 + 1)]  # duplicated triangles: the lower primID wins
    mesh = scenes.MeshData(v, t)
    ad, om = HipMeshAdapter(mesh), orc.Mesh(v, t)
    lo, hi = v[t.reshape(-1)].min(axis=0), v[t.reshape(-1)].max(axis=0)
    ext = np.maximum(hi - lo, 1e-6 * scale)
    n = int(rng.integers(1, 5000))
    org = (lo - 0.5 * ext + 2.0 * ext * rng.random((n, 3))).astype(np.float32)  # outside and inside the box
    tgt = v[t[rng.integers(0, n_t, n)]].astype(np.float64)
    w = rng.dirichlet([1.0, 1.0, 1.0], n)
    kind = rng.integers(0, 5, n)
    w[kind == 1] = [1.0, 0.0, 0.0]               # through a vertex
    w[kind == 2, 2] = 0.0                         # through an edge
    w[kind == 2] /= np.maximum(w[kind == 2].sum(axis=1, keepdims=True), 1e-9)
    p = (tgt * w[:, :, None]).sum(axis=1)
    d = p - org
    ax = kind == 3                                # axis-parallel: two zero components
    d[ax] = np.eye(3)[rng.integers(0, 3, ax.sum())] * rng.choice([-1.0, 1.0], (ax.sum(), 1))
    inpl = kind == 4                              # inside the plane of the target triangle
    d[inpl] = (tgt[inpl, 1] - tgt[inpl, 0]) + 1e-30
    nrm = np.linalg.norm(d, axis=1, keepdims=True)
    d = np.where(nrm > 0, d / np.maximum(nrm, 1e-300), [0.0, 0.0, 1.0]).astype(np.float32)
    # the reference is the DEFINITION: the arg-min of the triangle test over all triangles (the checker's brute-force loop), not the checker's own
    # tree -- where the test's t is noise (a ray through a vertex almost in the triangle's plane) a tree's culling order decides between duplicates
    # (every tree culls against the best hit with a relative slack of 2^-10 for that: gvt_device.h cull_bound.)  What no tree can return is a "hit" the
    # triangle test reports FAR from its triangle -- a ray almost in the plane of a triangle it passes at a distance: den ~ 0, t, u, v garbage;
    # the brute-force loop tests that triangle, a tree never enters its box (fuzz seed 295: a hit point 9 units outside the box of a 20-unit triangle
    # at t = 1,650).  Such rays are compared tree against tree; their number is bounded.
    brute, got, tree = om.intersect(org, d, use_bvh=False), ad.intersect(org, d), om.intersect(org, d)
    hp = org.astype(np.float64) + d.astype(np.float64) * brute["t"].astype(np.float64)[:, None]
    tv = v[t[np.maximum(brute["prim"], 0)]].astype(np.float64)
    off = np.maximum(tv.min(axis=1) - hp, hp - tv.max(axis=1)).max(axis=1)  # how far outside its triangle's box the brute-force hit point lies
    garbage = (brute["prim"] >= 0) & (off > np.abs(brute["t"]) * 2.0 ** -11)
    assert garbage.sum() <= max(8, n // 20)  # (a fifth of the rays is aimed INSIDE the plane of its target triangle)
    assert_hits_equal(got[~garbage], brute[~garbage])  # the device against the definition
    assert_hits_equal(tree[~garbage], brute[~garbage])  # the checker's own tree against the definition (its boxes are widened sideways by 2^-11 of the distance for this:
                                                         # oracle/gvt_oracle.c box_test; soak seeds 2044 / 3662 / 4948 / 8347)
    # garbage: whether a tree enters the far triangle's box is the tree's business -- the device returns the definition's hit or the checker tree's
    same = lambda x, y: (x["prim"] == y["prim"]) & (bits(x["t"]) == bits(y["t"])) & (bits(x["u"]) == bits(y["u"])) & (bits(x["v"]) == bits(y["v"]))  # noqa: E731
    g_ok = same(got, brute) | same(got, tree)
    # ... or, rarely (soak seeds 3358 / 4948 / 6715), a third garbage hit of the same ray that only its padded boxes reach: then it must be what the triangle test
    # says for the triangle it names, and not nearer than the definition's
    for i in np.nonzero(garbage & ~g_ok)[0]:
        assert got["prim"][i] >= 0 and got["t"][i] >= brute["t"][i], i
        one_tri = orc.Mesh(v, t[got["prim"][i]:got["prim"][i] + 1]).intersect(org[i:i + 1], d[i:i + 1], use_bvh=False)
        assert bits(one_tri["t"])[0] == bits(got["t"])[i] and bits(one_tri["u"])[0] == bits(got["u"])[i] and bits(one_tri["v"])[0] == bits(got["v"])[i], i
        g_ok[i] = True
    assert g_ok[garbage].all()
    occ = om.occluded(org, d, use_bvh=False)
    assert (ad.occluded(org, d)[~garbage] == occ[~garbage]).all() and (om.occluded(org, d)[~garbage] == occ[~garbage]).all()


def test_mesh_create_rejects_bad_input(hip):
    from gravit_amd import capi

    bad = scenes.MeshData(np.zeros((3, 3), np.float32), np.array([[0, 1, 7]], np.int32))
    with pytest.raises(capi.GvtHipError, match="references vertex"):
        HipMeshAdapter(bad)


def test_generated_normals_match_reference_hash(hip, ref_vectors):
    sc = scenes.bunny_scene()
    ad = HipMeshAdapter(sc.meshes[0])
    assert hashlib.sha256(ad.normals().tobytes()).hexdigest() == ref_vectors["bunny_normals_sha256"]


# ------------------------------------------------------------------ the adapter boundary: Adapter::trace
@pytest.mark.parametrize("mode", [NORMALS_FLAT, NORMALS_SMOOTH])
@pytest.mark.parametrize("name", ["bunny", "simple_cube", "soup"])
def test_trace_matches_oracle_bit_exact(hip, name, mode):
    if name == "bunny":
        sc, inst = scenes.bunny_scene(200, 150), 0
    elif name == "simple_cube":
        sc, inst = scenes.simple_scene(160, 160), 13
    else:
        sc, inst = scenes.soup_scene(100_000, 192, 108), 0
    mesh = sc.meshes[sc.inst_mesh[inst]]
    ad, om = HipMeshAdapter(mesh, mode), orc.Mesh(mesh.verts, mesh.tris, mesh_mat=mesh.material)
    rays = oracle_camera_rays(sc)
    # what the scheduler does before the adapter sees the rays: advance to the instance box
    nxt, t = orc.toplevel_intersect(sc.inst_lo, sc.inst_hi, orc.toplevel_order(sc.inst_lo, sc.inst_hi), rays)
    sel = nxt == inst
    rays = np.ascontiguousarray(rays[sel])
    rays["origin"] += rays["direction"] * (t[sel] * np.float32(0.95))[:, None]
    rg, rc = rays.copy(), rays.copy()
    out_g = ad.trace(rg, sc.m[inst], sc.minv[inst], sc.normi[inst], sc.lights)
    out_c = om.trace(rc, sc.m[inst], sc.minv[inst], sc.normi[inst], sc.lights, mode)
    assert len(out_g) == len(out_c) and len(out_c) > 0
    assert rays_equal_bits(sort_rays(out_g), sort_rays(out_c)), "moved_rays differ from the oracle"
    assert rays_equal_bits(rg, rc), "rayList was not updated in place like the oracle's"
    assert (out_g["type"] == 1).sum() > 0 and (out_g[out_g["type"] == 1]["t_max"] == np.float32(3.0)).all()


@pytest.mark.parametrize("seed", range(12))
def test_random_adapter_calls_against_the_oracle(hip, seed):
    """Seeded fuzz of Adapter::trace itself: a random mesh under a random translate / non-uniform scale (minv, normi as api.cpp:307-308
    makes them), one to three lights (point, ambient, area), rays of all three types with random depth, weight, colour and `t`, flat and
    smooth normals (generated or the caller's), Lambert / Phong / Blinn, per-vertex colours, per-face materials, a random [begin, end) range.  Moved rays as a set and the rayList updated in place, bit for
    bit (Lambert; the `powf` materials within 1e-5, counts exact)."""
    from gravit_amd.layouts import BLINN, LAMBERT, PHONG, ambient_light, area_light, default_material, point_light
    rng = np.random.default_rng(7000 + seed)
    n_v, n_t = int(rng.integers(30, 1500)), int(rng.integers(20, 4000))
    v = rng.normal(size=(n_v, 3)).astype(np.float32)
    t = rng.integers(0, n_v, (n_t, 3)).astype(np.int32)
    mtype = (LAMBERT, LAMBERT, PHONG, BLINN)[seed % 4]
    mat = default_material(kd=rng.uniform(0.1, 0.9, 3), mtype=mtype, ks=rng.uniform(0.1, 0.9, 3), alpha=float(rng.uniform(1.0, 30.0)))
    mode = NORMALS_SMOOTH if seed % 2 else NORMALS_FLAT
    vcol = rng.uniform(0, 1, (n_v, 3)).astype(np.float32) if seed % 6 == 3 else None            # per-vertex colours (a temporary Lambert per hit)
    mats, fmat = None, None
    if seed % 6 == 5:                                                                              # per-face materials, some faces without one (-1)
        mats = np.concatenate([default_material(kd=rng.uniform(0.1, 0.9, 3), mtype=LAMBERT) for _ in range(4)])
        fmat = rng.integers(-1, 4, n_t).astype(np.int32)
    vnrm = None
    if seed % 4 == 3:                                                                              # the caller's own vertex normals (not generated)
        vnrm = rng.normal(size=(n_v, 3)).astype(np.float32)
        vnrm /= np.linalg.norm(vnrm, axis=1, keepdims=True)
    mesh = scenes.MeshData(v, t, mat, vnormals=vnrm, vcolors=vcol, materials=mats, face_mat=fmat)
    ad, om = HipMeshAdapter(mesh, mode), orc.Mesh(v, t, vnormals=vnrm, vcolors=vcol, materials=mats, face_mat=fmat, mesh_mat=mat)
    m = scenes.mat_translate_scale(rng.uniform(-2, 2, 3), rng.uniform(0.3, 3.0, 3))
    minv, normi = scenes.instance_matrices(m)
    lights = [point_light(rng.uniform(-6, 6, 3), rng.uniform(0.2, 1.0, 3))]
    if seed % 3 == 1:
        lights.append(ambient_light(rng.uniform(0.05, 0.3, 3)))
    if seed % 3 == 2:
        lights.append(area_light(rng.uniform(-6, 6, 3), rng.uniform(0.2, 1.0, 3), (0.0, -1.0, 0.0), 0.7, 0.4))
        lights.append(point_light(rng.uniform(-6, 6, 3)))
    lights = np.concatenate(lights)
    n = int(rng.integers(1, 6000))
    rays = np.zeros(n, RAY_DTYPE)
    M = m.reshape(4, 4).T.astype(np.float64)
    world = v[t[rng.integers(0, n_t, n)]].astype(np.float64).mean(axis=1) @ M[:3, :3].T + M[:3, 3]
    rays["origin"] = (world + rng.normal(size=(n, 3)) * 4.0).astype(np.float32)
    d = world + rng.normal(size=(n, 3)) * 0.05 - rays["origin"]
    rays["direction"] = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    rays["t_min"], rays["t_max"], rays["t"] = RAY_EPSILON_F, np.float32(np.finfo(np.float32).max), rng.uniform(0.1, 5.0, n).astype(np.float32)
    rays["type"] = rng.choice([0, 0, 1, 2], n)
    rays["depth"], rays["w"], rays["id"] = rng.integers(1, 4, n), rng.uniform(0.05, 1.0, n).astype(np.float32), rng.integers(0, 1 << 20, n)
    rays["color"] = rng.uniform(0, 1, (n, 3)).astype(np.float32)
    rays["rng"] = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    begin = int(rng.integers(0, max(1, n // 3)))
    end = 0 if seed % 4 == 0 else int(rng.integers(begin, n + 1))
    rg, rc = rays.copy(), rays.copy()
    out_g = ad.trace(rg, m, minv, normi, lights, begin=begin, end=end, seed=seed)
    out_c = om.trace(rc, m, minv, normi, lights, mode, seed=seed, begin=begin, end=end)
    assert len(out_g) == len(out_c)
    if (end or n) - begin > 2000:  # the call did something: rays were hit (t updated in place), rays moved on
        assert (rg["t"] != rays["t"]).sum() > 20 and len(out_g) > 20
    a, b = sort_rays(out_g), sort_rays(out_c)
    if mtype == LAMBERT:
        assert rays_equal_bits(a, b) and rays_equal_bits(rg, rc)
    else:
        for f in RAY_DTYPE.names:
            if f == "color":
                assert np.abs(a[f] - b[f]).max(initial=0.0) <= 1e-5
            else:
                assert a[f].tobytes() == b[f].tobytes(), f
        assert rays_equal_bits(rg, rc)


def test_trace_ranges_capacity_and_errors(hip):
    from gravit_amd import capi
    import ctypes as C

    sc = scenes.bunny_scene(96, 96)
    mesh = sc.meshes[0]
    ad, om = HipMeshAdapter(mesh), orc.Mesh(mesh.verts, mesh.tris, mesh_mat=mesh.material)
    rays = oracle_camera_rays(sc)
    # begin/end sub-range (ragged), everything outside untouched
    rg, rc = rays.copy(), rays.copy()
    og = ad.trace(rg, sc.m[0], sc.minv[0], sc.normi[0], sc.lights, begin=777, end=5001)
    oc = om.trace(rc, sc.m[0], sc.minv[0], sc.normi[0], sc.lights, 0, begin=777, end=5001)
    assert rays_equal_bits(sort_rays(og), sort_rays(oc)) and rays_equal_bits(rg, rc)
    # empty list
    assert len(ad.trace(rays.copy()[:0], sc.m[0], sc.minv[0], sc.normi[0], sc.lights)) == 0
    # no lights: hits produce nothing, misses are forwarded
    r0 = rays.copy()
    o0 = ad.trace(r0, sc.m[0], sc.minv[0], sc.normi[0], sc.lights[:0])
    assert (o0["type"] == 0).all() and len(o0) == (r0["t"] == layouts.FLT_MAX).sum()
    # too small an output buffer is an error code + needed size, not a crash
    lib = capi.load()
    out = np.zeros(4, RAY_DTYPE)
    n_out = C.c_size_t(0)
    r1 = rays.copy()
    rc_ = lib.gvt_hip_trace(ad.h, capi.ptr(r1), C.c_size_t(len(r1)), C.c_size_t(0), C.c_size_t(0), capi.ptr(out), C.c_size_t(4), C.byref(n_out),
                            capi.ptr(capi.f32(sc.m[0])), capi.ptr(capi.f32(sc.minv[0])), capi.ptr(capi.f32(sc.normi[0])),
                            capi.ptr(np.ascontiguousarray(sc.lights, dtype=layouts.LIGHT_DTYPE)), C.c_size_t(1), C.c_int(0), C.c_uint32(0))
    assert rc_ == -3 and n_out.value == len(ad.trace(rays.copy(), sc.m[0], sc.minv[0], sc.normi[0], sc.lights))
    # bad arguments
    assert lib.gvt_hip_trace(ad.h, capi.ptr(r1), C.c_size_t(len(r1)), C.c_size_t(9), C.c_size_t(3), capi.ptr(out), C.c_size_t(4), C.byref(n_out),
                             capi.ptr(capi.f32(sc.m[0])), capi.ptr(capi.f32(sc.minv[0])), capi.ptr(capi.f32(sc.normi[0])),
                             capi.ptr(np.ascontiguousarray(sc.lights, dtype=layouts.LIGHT_DTYPE)), C.c_size_t(1), C.c_int(0), C.c_uint32(0)) == -1
    assert b"range" in lib.gvt_hip_last_error()
    # sizes beyond the 32-bit slot counters are refused before anything is allocated (a 65536 x 65536 film = 2^32 camera rays; a queue of 2^32 rays = 344 GB)
    q = RayQueue()
    with pytest.raises(capi.GvtHipError, match="32-bit"):
        q.reserve(1 << 32)
    q.append(rays[:10], keep_state=True)
    assert len(q) == 10  # the queue is as usable as before
    huge = scenes.Camera((0, 0, 3), (0, 0, 0), (0, 1, 0), 0.5, 65536, 65536)
    with pytest.raises(capi.GvtHipError, match="32-bit"):
        camera_generate(q, huge)
    top = TopLevel(sc.inst_lo, sc.inst_hi)
    pod = capi.CameraPod((C.c_float * 3)(0, 0, 3), (C.c_float * 3)(0, 0, 0), (C.c_float * 3)(0, 1, 0), 0.5, 65536, 65536, 1, 1, 0.0)
    qs = (C.c_void_p * 1)(q.h)
    assert lib.gvt_hip_camera_filter(top.h, C.byref(pod), C.c_int(8), qs, None) == -1 and b"2^32" in lib.gvt_hip_last_error()
    assert len(q) == 10 and len(ad.trace(rays.copy()[:100], sc.m[0], sc.minv[0], sc.normi[0], sc.lights)) > 0  # and the library goes on


def test_shadow_rays_that_hit_are_dropped_and_secondary_weight(hip):
    """Input lists with SHADOW and SECONDARY rays (EmbreeMeshAdapter.cpp:486-488, 572-575)."""
    sc = scenes.bunny_scene(64, 64)
    mesh = sc.meshes[0]
    ad, om = HipMeshAdapter(mesh), orc.Mesh(mesh.verts, mesh.tris, mesh_mat=mesh.material)
    rays = oracle_camera_rays(sc)
    rays["type"] = np.arange(len(rays)) % 3
    rg, rc = rays.copy(), rays.copy()
    og = ad.trace(rg, sc.m[0], sc.minv[0], sc.normi[0], sc.lights)
    oc = om.trace(rc, sc.m[0], sc.minv[0], sc.normi[0], sc.lights, 0)
    assert rays_equal_bits(sort_rays(og), sort_rays(oc)) and rays_equal_bits(rg, rc)


@pytest.mark.parametrize("mtype", [layouts.LAMBERT, layouts.PHONG, layouts.BLINN])
def test_materials_lights_vertex_colors(hip, mtype):
    """Phong/Blinn (powf), per-face materials, vertex colours, several lights incl. ambient and area."""
    sc = scenes.bunny_scene(96, 96)
    base = sc.meshes[0]
    rng = np.random.default_rng(4)
    mats = np.concatenate([layouts.default_material(kd=rng.random(3), mtype=mtype, ks=rng.random(3), alpha=1 + 4 * rng.random()) for _ in range(5)])
    face_mat = (np.arange(len(base.tris)) % 6 - 1).astype(np.int32)  # -1 -> mesh material
    lights = np.concatenate([layouts.point_light((0.0, 0.1, 0.5)), layouts.point_light((0.3, 0.4, 0.2), (0.2, 0.5, 0.9)),
                             layouts.ambient_light((0.05, 0.05, 0.1)), layouts.area_light((0.1, 0.5, 0.3), (1, 1, 1), (0.0, -1.0, 0.2), 0.2, 0.1)])
    for vcol in (None, rng.random(base.verts.shape).astype(np.float32)):
        mesh = scenes.MeshData(base.verts, base.tris, layouts.default_material(mtype=mtype), None, vcol, mats, face_mat)
        ad = HipMeshAdapter(mesh, NORMALS_SMOOTH)
        om = orc.Mesh(mesh.verts, mesh.tris, vcolors=vcol, materials=mats, face_mat=face_mat, mesh_mat=mesh.material)
        rays = oracle_camera_rays(sc)
        rg, rc = rays.copy(), rays.copy()
        og = sort_by_id_light(ad.trace(rg, sc.m[0], sc.minv[0], sc.normi[0], lights, seed=5))
        oc = sort_by_id_light(om.trace(rc, sc.m[0], sc.minv[0], sc.normi[0], lights, 1, seed=5))
        assert len(og) == len(oc)
        assert (og["id"] == oc["id"]).all() and (og["type"] == oc["type"]).all()
        assert (bits(og["origin"]) == bits(oc["origin"])).all() and (bits(og["direction"]) == bits(oc["direction"])).all()
        if mtype == layouts.LAMBERT or vcol is not None:  # vertex colours force LAMBERT (:556-557): no powf anywhere
            assert (bits(og["color"]) == bits(oc["color"])).all()
        else:
            assert np.abs(og["color"] - oc["color"]).max() <= RADIANCE_TOL
        assert rays_equal_bits(rg, rc)


@pytest.mark.parametrize("mtype", [layouts.EMBREE_METAL, layouts.EMBREE_VELVET, layouts.EMBREE_MATTE])
def test_embree_tutorial_brdfs(hip, mtype):
    """EMBREE_MATERIAL_METAL / VELVET / MATTE (Material.cpp:106-122, adapter/embree/EmbreeMaterial.h) through the adapter, as mesh
    material and as per-face materials: same shadow rays as the oracle (bit-exact origins / directions / ids), radiance to 1e-5
    (powf); the oracle itself is pinned on the reference's own build of these BRDFs (tests/test_oracle_pinning.py)."""
    sc = scenes.bunny_scene(96, 96)
    base = sc.meshes[0]
    rng = np.random.default_rng(40 + mtype)

    def mat():
        m = layouts.default_material(kd=rng.random(3), mtype=mtype, ks=rng.random(3))
        m["eta"] = 0.2 + 2.5 * rng.random(3)
        m["k"] = 1.0 + 5.0 * rng.random(3)
        m["roughness"] = 0.05 + 0.5 * rng.random()
        m["hsc"] = rng.random(3)
        m["backScattering"] = 0.2 + 1.5 * rng.random()
        m["hsFallOff"] = 1 + 9 * rng.random()
        return m

    mats = np.concatenate([mat() for _ in range(4)])
    face_mat = (np.arange(len(base.tris)) % 5 - 1).astype(np.int32)
    lights = np.concatenate([layouts.point_light((0.0, 0.1, 0.5)), layouts.point_light((0.3, 0.4, 0.2), (0.2, 0.5, 0.9))])
    mesh = scenes.MeshData(base.verts, base.tris, mat(), None, None, mats, face_mat)
    ad = HipMeshAdapter(mesh, NORMALS_SMOOTH)
    om = orc.Mesh(mesh.verts, mesh.tris, materials=mats, face_mat=face_mat, mesh_mat=mesh.material)
    rays = oracle_camera_rays(sc)
    rg, rc = rays.copy(), rays.copy()
    og = sort_by_id_light(ad.trace(rg, sc.m[0], sc.minv[0], sc.normi[0], lights, seed=5))
    oc = sort_by_id_light(om.trace(rc, sc.m[0], sc.minv[0], sc.normi[0], lights, 1, seed=5))
    assert len(og) == len(oc) and (og["type"] == 1).sum() > 1000
    assert (og["id"] == oc["id"]).all() and (og["type"] == oc["type"]).all()
    assert (bits(og["origin"]) == bits(oc["origin"])).all() and (bits(og["direction"]) == bits(oc["direction"])).all()
    assert np.abs(og["color"] - oc["color"]).max() <= RADIANCE_TOL
    assert og["color"][og["type"] == 1].max() > 0.05, "the lobes were never lit"
    assert rays_equal_bits(rg, rc)


def sort_by_id_light(r):
    """Order by (type, id, origin, direction): stable across tiny colour differences."""
    key = np.stack([r["type"], r["id"]] + [bits(r["origin"][:, k]).astype(np.int64) for k in range(3)] +
                   [bits(r["direction"][:, k]).astype(np.int64) for k in range(3)], 0)
    return r[np.lexsort(key[::-1])]


def test_bounce_math_device_bits_equal_host_bits(hip):
    """include/gvt_math.h on the device against the same header on the host, over EVERY value the reference's RNG can produce
    (RandEngine::rng returns k / 2^24, RandEngine.h:55): theta = (float)acos(sqrt(1 - Xi1)), phi = (float)(2 pi Xi2), and the sines
    and cosines of both -- the arithmetic CosWeightedRandomHemisphereDirection2 is built on (EmbreeMeshAdapter.cpp:289-318)."""
    xi = (np.arange(1 << 24, dtype=np.float64) / float(1 << 24)).astype(np.float32)
    theta_d, theta_h = hip.math_probe(2, xi), orc.math_probe(2, xi)
    assert (bits(theta_d) == bits(theta_h)).all()
    phi = (2.0 * 3.1415926535897932384626433832795 * xi.astype(np.float64)).astype(np.float32)
    for arg in (theta_h, phi, -phi[::97], phi[::89] * np.float32(100.0)):
        for kind in (0, 1):
            assert (bits(hip.math_probe(kind, arg)) == bits(orc.math_probe(kind, arg))).all()


def test_secondary_bounces_depth3(hip):
    """depth > 1: Russian roulette + cosine-weighted bounce (EmbreeMeshAdapter.cpp:584-602, 289-318) on the per-ray RNG
    streams.  sin / cos / acos are include/gvt_math.h on both sides, so the whole path is bit-exact: the in-place rayList (types,
    depths, origins, directions, weights, stream words) and the moved rays."""
    sc = scenes.cathedral_scene(64, 64, samples=1, depth=3)
    mesh = sc.meshes[0]
    ad, om = HipMeshAdapter(mesh), orc.Mesh(mesh.verts, mesh.tris, mesh_mat=mesh.material)
    rays = oracle_camera_rays(sc)
    rg, rc = rays.copy(), rays.copy()
    og = ad.trace(rg, sc.m[0], sc.minv[0], sc.normi[0], sc.lights, seed=11)
    oc = om.trace(rc, sc.m[0], sc.minv[0], sc.normi[0], sc.lights, 0, seed=11)
    assert (rc["type"] == 2).sum() > 100, "no bounces happened"
    assert rays_equal_bits(rg, rc), "rayList after the bounces differs from the oracle"
    assert len(og) == len(oc) and rays_equal_bits(sort_rays(og), sort_rays(oc))
    # the same with an area light in the scene: its sample positions come from the same per-ray stream (Light.cpp:115-127)
    lights = np.concatenate([sc.lights, layouts.area_light((0.0, 2.9, 2.0), (1, 1, 1), (0, -1, 0), 1.0, 1.0)])
    rg, rc = rays.copy(), rays.copy()
    og = ad.trace(rg, sc.m[0], sc.minv[0], sc.normi[0], lights, seed=3)
    oc = om.trace(rc, sc.m[0], sc.minv[0], sc.normi[0], lights, 0, seed=3)
    assert rays_equal_bits(rg, rc) and len(og) == len(oc) and rays_equal_bits(sort_rays(og), sort_rays(oc))


def test_device_born_rays_carry_their_rng_stream(hip):
    """Rays born on the device own their RNG stream (the word in Ray::data[64..67]): tracing a camera queue gives the same rays
    whatever the order of the list -- pixel-major like generateRays, or the 8x8 tiles the schedulers use."""
    sc = scenes.cathedral_scene(64, 48, samples=2, depth=3)
    ad = HipMeshAdapter(sc.meshes[0])
    outs = []
    for tile in (0, 8):
        q, mv = RayQueue(), RayQueue()
        camera_generate(q, sc.camera, tile)
        assert (q.to_numpy()["rng"] != 0).all()
        ad.trace_queue(q, mv, sc.m[0], sc.minv[0], sc.normi[0], sc.lights, seed=tile)
        outs.append(sort_rays(mv.to_numpy()))
    assert len(outs[0]) > 1000 and rays_equal_bits(outs[0], outs[1])
    cpu = oracle_camera_rays(sc)
    q = RayQueue()
    camera_generate(q, sc.camera, 0)
    assert rays_equal_bits(q.to_numpy(), cpu), "stream words of the camera rays differ from the oracle's"


@pytest.mark.parametrize("mode", [NORMALS_FLAT, NORMALS_SMOOTH])
def test_config5_cathedral_frame_image_scheduler(hip, mode):
    """BASELINE config 5 stand-in, whole frame: ~80 K long thin triangles, 1024x1024, samples = 2 (4 rays per pixel), depth 2,
    through the Image scheduler, against the oracle's restated scheduler.  Every ray is bit-exact (carried RNG streams, shared
    sin/cos/acos); the four deposits of a pixel meet in float atomics in arbitrary order -> max-abs <= 1e-5."""
    sc = scenes.cathedral_scene(1024, 1024, samples=2, depth=2, eye=(0.0, 1.5, 13.0), light=(0.0, 2.5, 12.0))
    tr = ImageTracer(sc, mode)
    B = tr()
    fb = B.framebuffer(True)
    ref, st = oracle_render(sc, mode, nthreads=8)
    assert st.rays_closest > 4 * 1024 * 1024 * 0.5 and st.rays_any > 1_000_000
    assert (ref[..., :3].sum(axis=2) > 0).mean() > 0.3, "the stand-in frame is mostly dark"
    assert np.abs(fb[..., :3] - ref[..., :3]).max() <= RADIANCE_TOL
    assert np.array_equal(fb[..., 3], ref[..., 3]), "deposit counts per pixel differ"  # every deposit adds exactly 1.0 to alpha
    g = hip.stats()
    assert tr.adapter_calls == st.adapter_calls


# ------------------------------------------------------------------ device queues, camera, top level, framebuffer
def test_queue_roundtrip_and_camera(hip):
    sc = scenes.simple_scene(130, 70)  # ragged sizes
    q = RayQueue()
    camera_generate(q, sc.camera)
    dev = q.to_numpy()
    cpu = oracle_camera_rays(sc)
    assert rays_equal_bits(dev, cpu), "camera rays differ from gvtPerspectiveCamera::generateRays restated"
    q2 = RayQueue(16)
    q2.append(cpu[:1000], keep_state=True)   # (rays in the library's own format: the oracle's camera rays carry their stream words)
    q2.append(cpu[1000:1003], keep_state=True)
    assert len(q2) == 1003 and rays_equal_bits(q2.to_numpy(), cpu[:1003])
    q2.clear()
    assert len(q2) == 0 and len(q2.to_numpy()) == 0
    cam2 = scenes.Camera((1, 2, 3), (0, 0.5, 0), (0.1, 1, 0), 0.7, 33, 17, samples=2, depth=2, jitter=1.0)
    camera_generate(q2, cam2)
    ref2 = orc.camera_rays(cam2.eye, cam2.focus, cam2.up, cam2.fov, 33, 17, 2, 2, 1.0)
    assert rays_equal_bits(q2.to_numpy(), ref2)
    # tiled listing (what the schedulers use): the same rays, 8x8 pixels at a time; ragged 33x17 exercises both edge strips
    camera_generate(q2, cam2, tile=8)
    t = q2.to_numpy()
    assert not rays_equal_bits(t, ref2)
    first_tile = t["id"][:64 * 4:4]  # samples=2 -> 4 rays per pixel
    assert sorted(first_tile.tolist()) == sorted((j * 33 + i) for j in range(8) for i in range(8))
    key = lambda r: np.lexsort((r["direction"][:, 2], r["direction"][:, 1], r["direction"][:, 0], r["id"]))
    assert rays_equal_bits(t[key(t)], ref2[key(ref2)])


def test_entry_points_no_other_test_calls(hip):
    """Four entries of include/gvt_hip.h that the bindings reach only through their wider siblings: gvt_hip_camera_generate (= _tiled with no
    tiles), gvt_hip_trace_queue (= _sink without a sink), gvt_hip_queue_sizes (n queues in one round trip) and gvt_hip_set_stream (a caller's
    hipStream_t instead of the context's own)."""
    import ctypes as C

    from gravit_amd import capi
    lib = capi.load()
    sc = scenes.bunny_scene(96, 64)
    cam = sc.camera
    qa, qb = RayQueue(), RayQueue()
    capi.check(lib.gvt_hip_camera_generate(qa.h, capi.ptr(capi.f32(cam.eye, 3)), capi.ptr(capi.f32(cam.focus, 3)), capi.ptr(capi.f32(cam.up, 3)), C.c_float(cam.fov),
                                           C.c_int(cam.width), C.c_int(cam.height), C.c_int(cam.samples), C.c_int(cam.depth), C.c_float(cam.jitter)), "gvt_hip_camera_generate")
    camera_generate(qb, cam, tile=0)
    rays = qa.to_numpy()
    assert len(rays) == 96 * 64 and rays_equal_bits(rays, qb.to_numpy()) and rays_equal_bits(rays, oracle_camera_rays(sc))
    # sizes of several queues at once
    qs = [qa, qb, RayQueue(), RayQueue(8)]
    qs[3].append(rays[:5], keep_state=True)
    arr = (C.c_void_p * len(qs))(*[q.h for q in qs])
    out = (C.c_uint64 * len(qs))()
    capi.check(lib.gvt_hip_queue_sizes(arr, C.c_size_t(len(qs)), out), "gvt_hip_queue_sizes")
    assert list(out) == [len(q) for q in qs] == [6144, 6144, 0, 5]
    # Adapter::trace on queues, without a sink, on the caller's stream; the same call with the (empty) sink on the context's own stream
    ad = HipMeshAdapter(sc.meshes[0], NORMALS_SMOOTH)
    f32 = lambda a, n: capi.ptr(capi.f32(a, n))  # noqa: E731
    lights = np.ascontiguousarray(sc.lights, dtype=layouts.LIGHT_DTYPE)
    hip_rt = C.CDLL("libamdhip64.so")  # the runtime the library itself is linked against (already in the process)
    st = C.c_void_p()
    assert hip_rt.hipStreamCreate(C.byref(st)) == 0 and st.value
    capi.set_stream(st.value)
    try:
        out_a = RayQueue()
        capi.check(lib.gvt_hip_trace_queue(ad.h, qa.h, out_a.h, f32(sc.m[0], 16), f32(sc.minv[0], 16), f32(sc.normi[0], 9), capi.ptr(lights), C.c_size_t(len(lights)),
                                           C.c_int(NORMALS_SMOOTH), C.c_uint32(0)), "gvt_hip_trace_queue")
        moved_a = out_a.to_numpy()
    finally:
        capi.set_stream(None)
        assert hip_rt.hipStreamDestroy(st) == 0
    out_b = RayQueue()
    ad.trace_queue(qb, out_b, sc.m[0], sc.minv[0], sc.normi[0], sc.lights, seed=0)
    moved_b = out_b.to_numpy()
    assert len(qa) == 0 and len(qb) == 0  # consumed, like ImageTracer.h:248
    ref = oracle_meshes(sc)[0].trace(oracle_camera_rays(sc), sc.m[0], sc.minv[0], sc.normi[0], sc.lights, NORMALS_SMOOTH)
    assert len(moved_a) > 1000 and rays_equal_bits(sort_rays(moved_a), sort_rays(moved_b)) and rays_equal_bits(sort_rays(moved_a), sort_rays(ref))
    ad.close()


def test_toplevel_shuffle_matches_oracle(hip):
    sc = scenes.simple_scene(150, 150)
    top = TopLevel(sc.inst_lo, sc.inst_hi)
    order = orc.toplevel_order(sc.inst_lo, sc.inst_hi)
    assert (top.order() == order).all()
    cpu = oracle_camera_rays(sc)
    q = RayQueue()
    q.append(cpu, keep_state=True)
    queues = [RayQueue() for _ in range(sc.n_inst)]
    fb = FrameBuffer(150, 150)
    top.shuffle(q, -1, queues, fb)
    assert len(q) == 0
    nxt, t = orc.toplevel_intersect(sc.inst_lo, sc.inst_hi, order, cpu)
    for i in range(sc.n_inst):
        exp = cpu[nxt == i].copy()
        exp["origin"] = exp["origin"] + exp["direction"] * (t[nxt == i] * np.float32(0.95))[:, None]
        got = queues[i].to_numpy()
        assert len(got) == len(exp) and rays_equal_bits(sort_rays(got), sort_rays(exp)), "queue %d" % i
    # second shuffle from an instance, with SHADOW rays: t_max=3 bounds the test, free ones deposit color*w
    shadow = cpu[:5000].copy()
    shadow["type"] = 1
    shadow["t_max"] = 3.0
    shadow["color"] = (0.25, 0.5, 0.75)
    shadow["w"] = 0.5
    q.append(shadow)
    for qq in queues:
        qq.clear()
    top.shuffle(q, 12, queues, fb)
    nxt2, _ = orc.toplevel_intersect(sc.inst_lo, sc.inst_hi, order, shadow, 12)
    assert [len(qq) for qq in queues] == [(nxt2 == i).sum() for i in range(sc.n_inst)]
    exp_fb = np.zeros((150 * 150, 4), np.float32)
    free = shadow[nxt2 < 0]
    np.add.at(exp_fb[:, :3], free["id"], free["color"] * free["w"][:, None])
    np.add.at(exp_fb[:, 3], free["id"], 1.0)
    assert np.array_equal(fb.download(False).reshape(-1, 4), exp_fb)
    # drop mode (shuffleDropRays): only kept queues receive rays
    q.append(cpu)
    for qq in queues:
        qq.clear()
    keep = np.array([i % 2 for i in range(sc.n_inst)], np.uint8)
    top.shuffle(q, -1, queues, None, keep)
    assert [len(qq) for qq in queues] == [int((nxt == i).sum()) if keep[i] else 0 for i in range(sc.n_inst)]


def test_shuffle_keeps_list_order_and_fused_camera_filter(hip):
    """The shuffle is order preserving and deterministic (<= 64 destinations): queue i holds exactly the rays of the input list whose
    next instance is i, in list order.  gvt_hip_camera_filter (generateRays fused into FilterRaysLocally) must produce the same queues
    as camera_generate + shuffle, for the pixel-major and the tiled listing, with and without a keep mask."""
    import ctypes as C
    from gravit_amd import capi

    sc = scenes.simple_scene(150, 130)
    top = TopLevel(sc.inst_lo, sc.inst_hi)
    order = orc.toplevel_order(sc.inst_lo, sc.inst_hi)
    cam = sc.camera
    pod = capi.CameraPod((C.c_float * 3)(*cam.eye), (C.c_float * 3)(*cam.focus), (C.c_float * 3)(*cam.up), cam.fov, cam.width, cam.height,
                         cam.samples, cam.depth, cam.jitter)
    for tile in (0, 8):
        for keep in (None, np.array([i % 3 != 0 for i in range(sc.n_inst)], np.uint8)):
            q = RayQueue()
            camera_generate(q, cam, tile=tile)
            lst = q.to_numpy()
            nxt, t = orc.toplevel_intersect(sc.inst_lo, sc.inst_hi, order, lst)
            qa = [RayQueue() for _ in range(sc.n_inst)]
            top.shuffle(q, -1, qa, None, keep)
            qb = [RayQueue() for _ in range(sc.n_inst)]
            arr = (C.c_void_p * sc.n_inst)(*[x.h for x in qb])
            capi.check(capi.load().gvt_hip_camera_filter(top.h, C.byref(pod), C.c_int(tile), arr, capi.ptr(keep)), "gvt_hip_camera_filter")
            for i in range(sc.n_inst):
                exp = lst[nxt == i].copy()  # list order
                exp["origin"] = exp["origin"] + exp["direction"] * (t[nxt == i] * np.float32(0.95))[:, None]
                if keep is not None and not keep[i]:
                    exp = exp[:0]
                a, b = qa[i].to_numpy(), qb[i].to_numpy()
                assert len(a) == len(exp) and rays_equal_bits(a, exp), "shuffle: queue %d not in list order" % i
                assert len(b) == len(exp) and rays_equal_bits(b, exp), "camera_filter: queue %d" % i


def test_long_ray_path_is_bit_exact(hip):
    """Rays that exceed `long_steps` node steps are parked by k_trace and finished by k_long_closest, a wave per ray.  Forced onto
    (nearly) every ray here -- threshold 1..24 steps, no minimum launch size; in the experiments build also without the saved stack and
    with a lower threshold for draining waves -- hits and whole frames must not change by a bit."""
    from gravit_amd import capi

    v, t = scenes.load_mesh_file(os.path.join(GOLDEN, "bun_zipper.npz"))
    mesh = scenes.MeshData(v, t, scenes.default_material())
    ad = HipMeshAdapter(mesh, NORMALS_SMOOTH)
    rng = np.random.default_rng(5)
    n = 30000
    lo, hi = v.min(0), v.max(0)
    org = (rng.uniform(-1, 2, (n, 3)) * (hi - lo) + lo).astype(np.float32)
    tgt = (rng.uniform(0, 1, (n, 3)) * (hi - lo) + lo).astype(np.float32)
    d = tgt - org
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    sc = scenes.bunny_grid_scene(width=190, height=108)
    try:
        capi.set_option("long_steps", 0)
        ref_hits = ad.intersect(org, d.astype(np.float32))
        ref_fb = ImageTracer(sc, NORMALS_SMOOTH)().framebuffer(False).copy()
        assert (ref_hits["prim"] >= 0).sum() > 1000
        capi.set_option("long_min_rays", 0)
        # long_save: the record carries the ray's pending stack (k_long_closest goes on from it) or not (it starts again at the root);
        # long_steps_drain: the lower threshold of waves whose work counter has run dry
        exp = capi.load().gvt_hip_is_experiments_build() == 1  # (tests/experiment_cases.py runs this test against the experiments build too)
        for steps, save, drain in ((1, 1, 0), (3, 1, 0), (8, 1, 0), (24, 1, 0)) + (((3, 0, 0), (8, 0, 0), (24, 1, 4), (24, 0, 2)) if exp else ()):
            capi.set_option("long_steps", steps)
            if exp:
                capi.set_option("long_save", save)
                capi.set_option("long_steps_drain", drain)
            h = ad.intersect(org, d.astype(np.float32))
            assert h.tobytes() == ref_hits.tobytes(), "hits differ with long_steps=%d long_save=%d long_steps_drain=%d" % (steps, save, drain)
            fb = ImageTracer(sc, NORMALS_SMOOTH)().framebuffer(False)
            assert np.array_equal(fb, ref_fb), "frame differs with long_steps=%d long_save=%d long_steps_drain=%d" % (steps, save, drain)
    finally:
        capi.set_option("defaults", 0)


def test_terminal_sink_equals_shuffle(hip):
    """gvt_hip_trace_queue_sink deposits terminal shadow rays inside the adapter; the frame must equal the one where every moved ray
    goes through shuffleRays (term_sink=0), on a scene where shadow rays do cross other instances."""
    from gravit_amd import capi

    sc = scenes.bunny_grid_scene(width=190, height=108)
    frames = {}
    for sink in (0, 1):
        capi.set_option("term_sink", sink)
        for native in (True, False):
            tr = ImageTracer(sc, NORMALS_SMOOTH, native=native)
            frames[(sink, native)] = (tr().framebuffer(False).copy(), tr.adapter_calls)
    capi.set_option("defaults", 0)
    ref_fb, ref_calls = frames[(0, True)]
    assert ref_fb[..., :3].sum() > 0
    for k, (fb, calls) in frames.items():
        assert calls == ref_calls and np.array_equal(fb, ref_fb), k


def test_framebuffer_clamp_and_ppm(hip):
    fb = FrameBuffer(8, 4)
    top = TopLevel(np.array([[10, 10, 10]], np.float32), np.array([[11, 11, 11]], np.float32))
    r = np.zeros(6, RAY_DTYPE)
    r["origin"] = (0, 0, 0)
    r["direction"] = (0, 0, 1)
    r["type"] = 1
    r["t_max"] = 3.0
    r["w"] = 1.0
    r["id"] = [0, 0, 0, 5, 31, 31]
    r["color"] = [(0.5, 0.25, 0.1)] * 3 + [(0.2, 0.4, 0.6)] + [(0.9, 0.9, 0.9)] * 2
    q = RayQueue()
    q.append(r)
    top.shuffle(q, -1, [RayQueue()], fb)
    raw, cl = fb.download(False), fb.download(True)
    assert np.allclose(raw[0, 0], (1.5, 0.75, 0.3, 3)) and np.allclose(cl[0, 0], (1.0, 0.75, 0.3, 3))
    assert np.allclose(cl[3, 7], (1.0, 1.0, 1.0, 2))
    assert (fb.ppm_bytes() == orc.fb_to_ppm_bytes(cl)).all()
    fb.clear()
    assert not fb.download(False).any()


# ------------------------------------------------------------------ whole frames
@pytest.mark.parametrize("name,builder", [("simple", scenes.simple_scene), ("bunny", scenes.bunny_scene)])
def test_reference_golden_images_on_gpu(hip, name, builder, oracle_vectors):
    """The reference's CTest on the HIP adapter: sum|byte diff| < 300 vs Test/CTESTtest/data/<name>.ppm (smooth mode),
    and the float framebuffer bit-identical to the oracle's in both normal modes."""
    sc = builder()
    for mode in (NORMALS_SMOOTH, NORMALS_FLAT):
        tr = ImageTracer(sc, mode)
        B = tr()
        fb = B.framebuffer(True)
        rec = oracle_vectors["fb_hashes"]["%s_mode%d" % (name, mode)]
        assert hashlib.sha256(np.ascontiguousarray(fb[..., :3]).tobytes()).hexdigest() == rec["rgb_sha256"]
        assert tr.adapter_calls == rec["adapter_calls"]
        if mode == NORMALS_SMOOTH:
            img = B.fb.ppm_bytes().astype(np.int64)
            gold = read_ppm(os.path.join(GOLDEN, "ref_%s.ppm" % name)).astype(np.int64)
            assert np.abs(img - gold).sum() < 300


def test_native_and_python_image_loops_agree(hip):
    """gvt_hip_image_frame (the loop inside the library) and the Python ImageTracer loop over the same entry points."""
    sc = scenes.simple_scene(256, 256)
    a = ImageTracer(sc, NORMALS_SMOOTH, native=True)
    b = ImageTracer(sc, NORMALS_SMOOTH, native=False)
    fa, fb_ = a().framebuffer(True).copy(), b().framebuffer(True)
    assert np.array_equal(fa, fb_) and a.adapter_calls == b.adapter_calls and a.adapter_calls >= 25


def test_config2_bunny70k_1080p(hip):
    """BASELINE config 2: bun_zipper (69,451 tris), 1920x1080, primary + shadow: integer framebuffer identical to the oracle."""
    sc = scenes.bunny70k_scene()
    B = ImageTracer(sc, NORMALS_FLAT)()
    fb = B.framebuffer(True)
    ref, st = oracle_render(sc, 0, nthreads=8)
    assert np.array_equal(fb[..., :3], ref[..., :3])
    assert (B.fb.ppm_bytes() == orc.fb_to_ppm_bytes(ref)).all()
    assert st.rays_closest > 400_000 and st.rays_any > 300_000


def test_config4_bunny_grid_image_scheduler(hip):
    """BASELINE config 4 geometry on one GPU (8 instances of one mesh, one cached adapter): equals the oracle."""
    sc = scenes.bunny_grid_scene(width=475, height=270)
    tr = ImageTracer(sc, NORMALS_SMOOTH)
    B = tr()
    ref, st = oracle_render(sc, 1)
    assert np.array_equal(B.framebuffer(True)[..., :3], ref[..., :3])
    assert tr.adapter_calls == st.adapter_calls and len(B.adapter_cache) == 1


def test_many_instances_take_the_unordered_shuffle(hip):
    """More than 64 destinations: the shuffle falls back from the scan-ordered slots to per-block atomics (queue order then is
    arrival order, which the reference does not define either).  96 bunny instances, image and adapter-call count equal the oracle."""
    sc = scenes.bunny_grid_scene(nx=12, ny=8, pitch=0.22, width=240, height=160)
    assert sc.n_inst == 96
    tr = ImageTracer(sc, NORMALS_SMOOTH)
    B = tr()
    ref, st = oracle_render(sc, 1)
    assert np.array_equal(B.framebuffer(True)[..., :3], ref[..., :3])
    assert tr.adapter_calls == st.adapter_calls and (ref[..., :3].sum(axis=2) > 0).sum() > 1000


def test_legacy_conf_scene_three_bunnies(hip):
    """BASELINE config 1/4 input: data/bunny.conf (3 instances of bunny.obj, Domain scheduler hint), reduced film."""
    sc = scenes.load_conf(os.path.join(GOLDEN, "bunny.conf"), width=475, height=270)
    tr = ImageTracer(sc, NORMALS_SMOOTH)
    B = tr()
    ref, st = oracle_render(sc, 1)
    assert np.array_equal(B.framebuffer(True)[..., :3], ref[..., :3]) and tr.adapter_calls == st.adapter_calls
    assert (ref[..., :3].sum(axis=2) > 0).sum() > 2000


def test_multisample_frame_within_tolerance(hip):
    """samples=2: four contributions per pixel arrive through float atomics in arbitrary order -> 1e-5, not bit-exact."""
    sc = scenes.bunny_scene(128, 128)
    sc.camera.samples = 2
    B = ImageTracer(sc, NORMALS_SMOOTH)()
    ref, _ = oracle_render(sc, 1)
    assert np.abs(B.framebuffer(True)[..., :3] - ref[..., :3]).max() <= RADIANCE_TOL


def test_trace_without_write_back_leaves_the_ray_list_alone(hip):
    """GVT_HIP_TRACE_NO_WRITEBACK (gvt_hip_trace_ex): the same moved rays, rayList untouched -- what the reference's schedulers need,
    which clear the traced queue right after Adapter::trace (ImageTracer.h:248, DomainTracer.h:316)."""
    sc = scenes.bunny_grid_scene(width=96, height=54)
    ad = HipMeshAdapter(sc.meshes[sc.inst_mesh[0]], NORMALS_SMOOTH)
    c = sc.camera
    rays = orc.camera_rays(c.eye, c.focus, c.up, c.fov, c.width, c.height)
    a, b = rays.copy(), rays.copy()
    moved_a = ad.trace(a, sc.m[0], sc.minv[0], sc.normi[0], sc.lights, seed=3)
    moved_b = ad.trace(b, sc.m[0], sc.minv[0], sc.normi[0], sc.lights, seed=3, write_back=False)
    assert rays_equal_bits(b, rays) and not rays_equal_bits(a, rays)  # the 68 defined bytes of every ray (numpy leaves the padding of a copy undefined)
    assert len(moved_a) == len(moved_b) and len(moved_a) > 0
    assert rays_equal_bits(sort_rays(moved_a), sort_rays(moved_b))


def test_simd_cpu_baseline_equals_the_oracle_on_the_downloaded_tree(hip):
    """bench.py times a SIMD CPU baseline beside the GPU number (oracle/simd_baseline.c: one ray against four quantised boxes per step over the
    GPU-built 4-wide tree).  Its answers must be the oracle's, bit for bit, before its time means anything: closest hits and occlusion flags on
    the bunny and on a soup, rays through the box, along the axes and from inside."""
    from oracle import simd

    from oracle import orc

    for name, mesh in (("bunny", scenes.bunny_scene().meshes[0]), ("soup", scenes.soup_scene(200_000, 64, 64).meshes[0])):
        om = orc.Mesh(mesh.verts, mesh.tris)
        ad = HipMeshAdapter(mesh)
        nodes4, slots = ad.download_wide()
        assert nodes4.shape[1] == 16 and slots.shape == (len(mesh.tris), 16) and len(nodes4) > 0
        T = simd.Tree(nodes4, slots)
        lo, hi = om.bbox()
        org, d = seeded_rays_at(lo, hi, 30_011, 5)
        ref = om.intersect(org, d)
        t, prim, u, v = T.intersect(org, d, nthreads=4)
        assert np.array_equal(ref["prim"], prim) and np.array_equal(bits(ref["t"]), bits(t)) and np.array_equal(bits(ref["u"]), bits(u)) and np.array_equal(bits(ref["v"]), bits(v)), name
        assert (prim >= 0).sum() > 1000
        assert np.array_equal(om.occluded(org, d), T.occluded(org, d, nthreads=4)), name
        assert_hits_equal(ad.intersect(org, d), ref)
        if simd.load8() is not None:  # the 8-wide baseline (AVX2: one ray against eight boxes of the tree collapsed once more on the host): the same answers
            T8 = simd.Tree8(nodes4, slots)
            assert 0 < T8.n8 < T8.n4 and 4.0 < T8.children_per_node <= 8.0, (name, T8.n8, T8.n4, T8.children_per_node)
            t8, prim8, u8, v8 = T8.intersect(org, d, nthreads=4)
            assert np.array_equal(ref["prim"], prim8) and np.array_equal(bits(ref["t"]), bits(t8)) and np.array_equal(bits(ref["u"]), bits(u8)) and np.array_equal(bits(ref["v"]), bits(v8)), name
            assert np.array_equal(om.occluded(org, d), T8.occluded(org, d, nthreads=4)), name
            assert T8.last_steps[0] > 0


def test_host_appended_rays_start_fresh_whatever_their_padding_holds(hip):
    """gvt_hip_queue_append without GVT_HIP_APPEND_KEEP_STATE: bytes 64..79 of a host ray are NOT read as scheduler state.  The reference's
    Ray(origin, dir, ...) constructor never writes data[64..79] (Ray.h:106-116), so a GraviT host hands over stack garbage there; read as a
    known-miss list, a 16-bit entry equal to an instance number + 1 would let shuffleRays walk the ray through that instance without tracing
    it.  Camera rays of a scene of overlapping instances with every such entry poisoned (all instances 'already crossed') are shuffled
    exactly like clean ones; with the flag the same bytes ARE state and the shuffle walks through the boxes."""
    hip.set_option("skip_known", 1)  # the shortcut on: the setting under which a known-miss list is read at all
    sc = scenes.simple_scene(96, 96)
    top = TopLevel(sc.inst_lo, sc.inst_hi)
    clean = oracle_camera_rays(sc)
    clean["rng"] = 0
    clean["known"] = 0
    dirty = clean.copy()
    dirty["rng"] = 0xDEADBEEF
    dirty["known"] = np.arange(1, 7, dtype=np.uint16)[None, :]  # instances 0..5 "known missed"

    def shuffled(rays, keep):
        q = RayQueue()
        q.append(rays, keep_state=keep)
        queues = [RayQueue() for _ in range(sc.n_inst)]
        top.shuffle(q, 12, queues, FrameBuffer(96, 96))  # as rays that leave instance 12 (the camera's own shuffle, from -1, reads no list)
        return [qq.to_numpy() for qq in queues]

    a, b = shuffled(clean, False), shuffled(dirty, False)
    assert sum(len(x) for x in a) > 1000
    for x, y in zip(a, b):
        assert len(x) == len(y) and np.array_equal(sort_rays(x).view(np.uint8).reshape(-1, 80), sort_rays(y).view(np.uint8).reshape(-1, 80))
    c = shuffled(dirty, True)  # the same bytes taken as state: rays bound for instances 0..5 are walked past them
    assert [len(x) for x in c] != [len(x) for x in a]
    # the entry point of the earlier ABI revisions keeps its boolean: 0 = host rays (fresh), 1 = a device wire image (state kept); a flag word is refused
    import ctypes as C
    lib = hip.load()
    q = RayQueue()
    d = np.ascontiguousarray(dirty[:256])
    assert lib.gvt_hip_queue_append(q.h, hip.ptr(d), C.c_size_t(len(d)), C.c_int(2)) == -1 and "gvt_hip_queue_append_flags" in hip.last_error()
    assert lib.gvt_hip_queue_append(q.h, hip.ptr(d), C.c_size_t(len(d)), C.c_int(0)) == 0
    got = q.to_numpy()
    assert len(got) == 256 and not got["known"].any() and np.array_equal(got["origin"], d["origin"])


@pytest.mark.parametrize("case", ["soup-300k", "bun_zipper", "clustered"])
def test_builder_node_boxes_in_lds_equal_the_range_union_table(hip, case, monkeypatch):
    """The builder's inner-node boxes come from two routes: nodes whose Karras range lies inside one chunk of 1,024 sorted triangles are unioned bottom-up in LDS
    (k_node_boxes_chunk), the spines across chunk boundaries from the base-32 range-union table (k_node_boxes_spine).  Forcing the table for EVERY node
    (GVT_HIP_NODE_BOXES_TABLE, the round-5 path) must give the same tree bit for bit -- every binary node with both child boxes, and the triangle slots -- on a soup, a
    surface and a mesh with everything crowded into a corner (many equal Morton keys: the tie-break's deep, lopsided subtrees)."""
    if case == "soup-300k":
        v, t = scenes.triangle_soup(300_000, seed=7)
    elif case == "bun_zipper":
        z = np.load(os.path.join(GOLDEN, "bun_zipper.npz"))
        v, t = z["verts"], z["tris"]
    else:
        rng = np.random.default_rng(3)
        c = np.concatenate([rng.random((40_000, 1, 3)) * 1e-4, rng.random((5_000, 1, 3))]).astype(np.float32)  # 40 K triangles in a 1e-4 corner of the unit box
        v = (c + (rng.random((len(c), 3, 3)).astype(np.float32) - 0.5) * np.float32(2e-5)).reshape(-1, 3)
        t = np.arange(len(c) * 3, dtype=np.int32).reshape(-1, 3)
    m = scenes.MeshData(np.ascontiguousarray(v, np.float32), np.ascontiguousarray(t, np.int32))
    a = HipMeshAdapter(m)
    na, (wa, sa) = a.download_nodes(), a.download_wide()
    monkeypatch.setenv("GVT_HIP_NODE_BOXES_TABLE", "1")
    b = HipMeshAdapter(m)
    nb, (wb, sb) = b.download_nodes(), b.download_wide()
    assert len(na) == len(nb) > 1000 and np.array_equal(na.view(np.uint32), nb.view(np.uint32))
    # (the 4-wide collapse numbers a level's nodes in block-arrival order: its array is a permutation from build to build; the slots are not)
    assert wa.shape == wb.shape and np.array_equal(sa.view(np.uint32), sb.view(np.uint32))
    a.close(); b.close()


@pytest.mark.parametrize("case", ["soup-300k", "bun_zipper", "one-leaf"])
def test_cluster_layout_is_the_same_tree(hip, case):
    """lbvh.hip build_nodes4c (round 6): the layout k_finish walks two levels per memory round trip.  Walked from the root's entry, level pair by level pair,
    beside the 4-wide nodes it was made from: every node appears once, with the same boxes; leaves keep their references; an even-level node is followed by
    its inner children in child order; the references of those to inner nodes are (slot << 4) | the mask of that node's inner children."""
    if case == "soup-300k":
        v, t = scenes.triangle_soup(300_000, seed=11)
    elif case == "bun_zipper":
        z = np.load(os.path.join(GOLDEN, "bun_zipper.npz"))
        v, t = z["verts"], z["tris"]
    else:
        v, t = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32), np.array([[0, 1, 2]], np.int32)
    a = HipMeshAdapter(scenes.MeshData(np.ascontiguousarray(v, np.float32), np.ascontiguousarray(t, np.int32)))
    w, _ = a.download_wide()
    c, root = a.download_clusters()
    assert c is not None and c.shape == w.shape and root >> 4 == 0
    REF = [10, 11, 12, 13]  # the four child references' columns (gvt_hip.h gvt_hip_mesh_download_wide)
    BOX = [i for i in range(16) if i not in REF]
    refs_w, refs_c = w[:, REF].view(np.int32), c[:, REF].view(np.int32)
    def inner_mask(r):  # (n, 4) int32 references -> bit c: child c is an inner node
        return ((r >= 0) * (1 << np.arange(4))).sum(1).astype(np.int64)
    seen = np.zeros(len(c), np.int64)
    wi, slot, mask = np.array([0], np.int64), np.array([root >> 4], np.int64), np.array([root & 15], np.int64)  # pairs (node of nodes4, its cluster's entry)
    levels = 0
    while len(wi):
        assert np.array_equal(w[wi][:, BOX], c[slot][:, BOX]) and np.array_equal(refs_w[wi] < 0, refs_c[slot] < 0)
        assert np.array_equal(mask, inner_mask(refs_w[wi]))
        leaf = refs_w[wi] < 0
        assert np.array_equal(refs_w[wi][leaf], refs_c[slot][leaf])
        np.add.at(seen, slot, 1)
        # the inner children: slot + 1 + (inner children before it)
        before = np.cumsum(~leaf, 1) - (~leaf)
        k, ch = np.nonzero(~leaf)
        cw, cs = refs_w[wi][k, ch].astype(np.int64), slot[k] + 1 + before[k, ch]
        assert np.array_equal(w[cw][:, BOX], c[cs][:, BOX])
        np.add.at(seen, cs, 1)
        rw, rc = refs_w[cw], refs_c[cs]
        gleaf = rw < 0
        assert np.array_equal(gleaf, rc < 0) and np.array_equal(rw[gleaf], rc[gleaf])
        k2, ch2 = np.nonzero(~gleaf)
        wi, ent = rw[k2, ch2].astype(np.int64), rc[k2, ch2].astype(np.int64)
        slot, mask = ent >> 4, ent & 15
        levels += 2
    assert np.all(seen == 1), "%d of %d cluster slots not reached exactly once" % (int((seen != 1).sum()), len(seen))
    assert (case == "one-leaf") == (len(c) == 1) and levels <= 32
    a.close()
