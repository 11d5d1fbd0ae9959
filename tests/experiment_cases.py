"""The variants that were measured and lost (EXPERIMENTS.md) and the tuned constants of the shipped kernels still have to return the oracle's bits whatever their setting: this module runs against the
EXPERIMENTS build of the library (libgvt_hip_exp.so, -DGVT_EXPERIMENTS), where they sit behind their knobs -- the first-version
kernels (trav_kernel=0), the binary-node and quad-cooperative-fetch arms of k_trace (wide4=0, coop_fetch=1), k_fused, k_packet and
k_traceq (four lanes per ray, quad=1, with its own node / leaf-block layouts).  It is not collected by `pytest tests` (the shipped
library refuses those knobs); tests/test_gpu_experiments.py starts it ONCE, in one child process, with GVT_HIP_LIB pointing at the
experiments build."""
import numpy as np
import pytest

from gravit_amd import capi, scenes
from gravit_amd.adapter import HipMeshAdapter
from gravit_amd.layouts import NORMALS_FLAT, NORMALS_SMOOTH
from gravit_amd.scheduler import NativeTracer
from oracle import orc
from tests.helpers import bits, oracle_camera_rays, oracle_render, rays_equal_bits, seeded_rays_at, sort_rays
from tests.test_gpu_native import config5
from tests.test_gpu_parity import assert_hits_equal

pytestmark = pytest.mark.gpu


def test_this_is_the_experiments_build(hip):
    assert capi.load().gvt_hip_is_experiments_build() == 1


@pytest.mark.parametrize("opts", [dict(refill_min=1, inner_min=1), dict(refill_min=64, inner_min=64), dict(blocks_per_cu=1, refill_min=8, inner_min=16), dict(sort_rays=0, top_lds=0),
                                  dict(sort_rays=1, sort_bits=32), dict(refill_min=2, inner_min=60), dict(share=0), dict(share=3, blocks_per_cu=6, refill_min=64, share_min_rays=0),
                                  dict(share=3, share_min_rays=0), dict(share=3, share_min_rays=0, long_steps=3, long_min_rays=0), dict(share=1, share_min_rays=0, blocks_per_cu=1),
                                  dict(top_ordered=0), dict(leaf_max=3, share=3, share_min_rays=0), dict(sort_rays=1, sort_gather=1), dict(term_sink=0, camera_tile=0),
                                  dict(trav_kernel=0), dict(wide4=0, coop_fetch=0), dict(wide4=0, coop_fetch=1, refill_min=3, inner_min=5), dict(wide4=0, sort_rays=1),
                                  dict(wide4=0, share=3, share_min_rays=0, long_steps=3, long_min_rays=0),
                                  dict(quad=1), dict(quad=1, leaf_max=4), dict(quad=1, leaf_max=1, quad_inner_min=1, quad_refill_min=1),
                                  dict(quad=1, leaf_max=3, quad_inner_min=16, quad_refill_min=16, blocks_per_cu_quad=1),
                                  dict(quad=1, leaf_max=4, long_steps=2, long_min_rays=0), dict(quad=1, leaf_max=4, long_steps=5, long_min_rays=0, long_save=0)])
def test_results_do_not_depend_on_experimental_variants(hip, opts):
    sc = scenes.soup_scene(150_000, 160, 90)
    mesh = sc.meshes[0]
    om = orc.Mesh(mesh.verts, mesh.tris, mesh_mat=mesh.material)
    lo, hi = om.bbox()
    org, d = seeded_rays_at(lo, hi, 30_001, 21)
    rays = oracle_camera_rays(sc)
    try:
        for k, v in opts.items():
            hip.set_option(k, v)
        ad = HipMeshAdapter(mesh)  # after the options: quad / leaf_max decide the layouts the mesh is built with
        assert_hits_equal(ad.intersect(org, d), om.intersect(org, d))
        assert (ad.occluded(org, d) == om.occluded(org, d)).all()
        rg, rc = rays.copy(), rays.copy()
        og = ad.trace(rg, sc.m[0], sc.minv[0], sc.normi[0], sc.lights)
        oc = om.trace(rc, sc.m[0], sc.minv[0], sc.normi[0], sc.lights, 0)
        assert rays_equal_bits(sort_rays(og), sort_rays(oc)) and rays_equal_bits(rg, rc)
    finally:
        hip.set_option("defaults", 0)


@pytest.mark.parametrize("name", ["bunny", "cube", "cathedral", "tiny"])
def test_quad_kernel_on_surface_meshes_and_deep_stacks(hip, name):
    """k_traceq on meshes with shared vertices, axis-aligned zero-thickness boxes, long thin triangles (deep stacks: the spill
    path) and a mesh smaller than one leaf."""
    mesh = {"bunny": lambda: scenes.bunny_scene().meshes[0], "cube": lambda: scenes.simple_scene().meshes[1],
            "cathedral": lambda: scenes.cathedral_scene(32, 32).meshes[0], "tiny": lambda: scenes.simple_scene().meshes[1]}[name]()
    if name == "tiny":
        mesh = scenes.MeshData(verts=mesh.verts, tris=np.ascontiguousarray(mesh.tris[:3]), material=mesh.material)
    om = orc.Mesh(mesh.verts, mesh.tris)
    lo, hi = om.bbox()
    org, d = seeded_rays_at(lo, hi, 20011, 3)
    if name == "cathedral":
        rng = np.random.default_rng(9)
        org = np.tile(np.array([0.0, 1.5, 5.0], np.float32), (20011, 1))
        d = rng.normal(size=(20011, 3)).astype(np.float32)
        d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    try:
        hip.set_option("quad", 1)
        hip.set_option("leaf_max", 4)
        ad = HipMeshAdapter(mesh)
        assert_hits_equal(ad.intersect(org, d), om.intersect(org, d))
        assert (ad.occluded(org, d) == om.occluded(org, d)).all()
    finally:
        hip.set_option("defaults", 0)


@pytest.mark.parametrize("opts", [dict(first_round_async=0), dict(wave_single=0), dict(shadow_direct=0), dict(small_rays=1 << 30, wave_single=0), dict(blocks_per_cu_closest=0, term_sink=0),
                                  dict(lean_frame=0), dict(report_poll=0), dict(lean_frame=0, report_poll=0, first_round_async=0), dict(sort_rays=1, camera_tile=0),
                                  dict(packet=2), dict(packet=2, camera_tile=0), dict(first_round_async=1, packet=2), dict(fused=1), dict(fused1=1, fused1_min_rays=0), dict(fused1=1, fused1_min_rays=0, packet=0), dict(quad=1, leaf_max=4),
                                  dict(quad=1, leaf_max=2, small_rays=0), dict(quad=1, leaf_max=4, wave_single=0, shadow_direct=0)])
def test_round_results_do_not_depend_on_experimental_variants(hip, opts):
    for sc, mode, tol in ((config5(192, 4), NORMALS_FLAT, 1e-5), (scenes.bunny_grid_scene(width=380, height=216), NORMALS_SMOOTH, 0.0),
                          (scenes.soup_scene(100_000, 160, 90), NORMALS_FLAT, 0.0), (scenes.bunny_scene(256, 256), NORMALS_SMOOTH, 0.0)):
        ref, st = oracle_render(sc, mode, nthreads=8)
        try:
            for k, v in opts.items():
                hip.set_option(k, v)
            tr = NativeTracer(sc, mode)
            fb = tr().framebuffer(True)
            assert np.abs(fb[..., :3] - ref[..., :3]).max() <= tol and np.array_equal(fb[..., 3], ref[..., 3])
            assert tr.stats["rays_closest"] == st.rays_closest and tr.stats["rays_any"] == st.rays_any
            tr.close()
        finally:
            hip.set_option("defaults", 0)


def test_binary_tree_and_quad_kernel_agree_with_the_default_at_one_million_triangles(hip):
    sc = scenes.soup_scene(1_000_000, 640, 360)
    rays = oracle_camera_rays(sc)
    o, d = rays["origin"], rays["direction"]
    a = HipMeshAdapter(sc.meshes[0]).intersect(o, d)
    try:
        hip.set_option("wide4", 0); hip.set_option("trav_kernel", 0)
        b = HipMeshAdapter(sc.meshes[0]).intersect(o, d)
        hip.set_option("defaults", 0)
        hip.set_option("quad", 1); hip.set_option("leaf_max", 4)
        c = HipMeshAdapter(sc.meshes[0]).intersect(o, d)
    finally:
        hip.set_option("defaults", 0)
    assert a.tobytes() == b.tobytes(), "4-wide compressed layout and binary tree disagree"
    assert a.tobytes() == c.tobytes(), "one lane per ray and four lanes per ray disagree"
    assert (a["prim"] >= 0).sum() > 50_000


def test_long_ray_path_variants(hip):
    """tests/test_gpu_parity.py::test_long_ray_path_is_bit_exact with the knobs only this build can move: no saved stack (long_save = 0),
    a lower parking threshold for draining waves (long_steps_drain)."""
    from tests.test_gpu_parity import test_long_ray_path_is_bit_exact

    test_long_ray_path_is_bit_exact(hip)
