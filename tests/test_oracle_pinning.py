"""Pins the CPU oracle: against the reference's golden images, against known answers produced by the
reference's OWN code (tests/golden/ref_vectors.json, generated through oracle/_ref by
tests/golden/make_fixtures.py) and -- when the reference build is present (build container) -- live.
No GPU needed."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

from gravit_amd import scenes
from oracle import orc
from tests.conftest import GOLDEN, read_ppm
from tests.helpers import oracle_meshes, oracle_render, oracle_render_domain

REF_TOLERANCE = 300  # gvtImageDiff -tolerance 300 (reference CMakeLists.txt:666, ImageDiff.cpp:45-66,97)


def test_struct_sizes_match_reference(ref_vectors):
    assert ref_vectors["sizes"] == {"ray": 80, "material": 92, "box3d": 32}
    assert orc.RAY_DTYPE.itemsize == 80 and orc.MATERIAL_DTYPE.itemsize == 92 and orc.LIGHT_DTYPE.itemsize == 64
    assert np.float32(ref_vectors["ray_epsilon"]) == np.float32(1e-6)
    # Material() initialises type..roughness (bytes 0-71); the velvet fields are left indeterminate (Material.h:62-77)
    assert orc.default_material().tobytes().hex()[:144] == ref_vectors["default_material_hex"][:144]


@pytest.mark.parametrize("name,builder", [("simple", scenes.simple_scene), ("bunny", scenes.bunny_scene)])
def test_reference_golden_images_smooth(name, builder):
    """The reference's CTest: render, diff against Test/CTESTtest/data/<name>.ppm, sum |byte diff| < 300.
    The goldens pre-date FLAT_SHADING: smooth normals reproduce them."""
    sc = builder()
    fb, st = oracle_render(sc, 1)
    img = orc.fb_to_ppm_bytes(fb).astype(np.int64)
    gold = read_ppm(os.path.join(GOLDEN, "ref_%s.ppm" % name)).astype(np.int64)
    d = np.abs(img - gold)
    assert d.sum() < REF_TOLERANCE, "sum-abs %d (%d px)" % (d.sum(), (d.sum(axis=2) > 0).sum())
    if name == "simple":
        assert d.sum() == 0  # reproduced exactly
        assert st.adapter_calls == 37  # 25 instances, queues drained in 37 adapter calls
    else:
        assert d.sum() <= 138 and (d.sum(axis=2) > 0).sum() <= 4


@pytest.mark.parametrize("name,builder", [("simple", scenes.simple_scene), ("bunny", scenes.bunny_scene)])
def test_flat_mode_is_not_the_golden(name, builder):
    """Flat normals (the current EmbreeMeshAdapter.cpp, FLAT_SHADING) differ from the goldens by >1e6: both modes exist."""
    fb, _ = oracle_render(builder(), 0)
    img = orc.fb_to_ppm_bytes(fb).astype(np.int64)
    gold = read_ppm(os.path.join(GOLDEN, "ref_%s.ppm" % name)).astype(np.int64)
    assert np.abs(img - gold).sum() > 1_000_000


def test_framebuffer_hashes_are_stable(oracle_vectors):
    for name, builder in (("simple", scenes.simple_scene), ("bunny", scenes.bunny_scene)):
        for mode in (0, 1):
            fb, st = oracle_render(builder(), mode)
            rec = oracle_vectors["fb_hashes"]["%s_mode%d" % (name, mode)]
            assert hashlib.sha256(np.ascontiguousarray(fb[..., :3]).tobytes()).hexdigest() == rec["rgb_sha256"]
            assert st.adapter_calls == rec["adapter_calls"] and st.rays_closest == rec["rays_closest"] and st.rays_any == rec["rays_any"]


def test_domain_scheduler_two_ranks_matches_golden_and_image_scheduler():
    """The reference runs the same goldens with -domain on 2 ranks (CMakeLists.txt:650-654, same tolerance)."""
    sc = scenes.simple_scene()
    owner = [(0 if sc.inst_mesh[i] == 0 else 1) for i in range(sc.n_inst)]  # SimpleApp.cpp:128-135: cones on even, cubes on odd ranks
    fb, st = oracle_render_domain(sc, owner, 2, 1)
    img = orc.fb_to_ppm_bytes(fb).astype(np.int64)
    gold = read_ppm(os.path.join(GOLDEN, "ref_simple.ppm")).astype(np.int64)
    assert np.abs(img - gold).sum() < REF_TOLERANCE
    assert st.rays_sent > 0 and st.rounds >= 2
    fb1, _ = oracle_render(sc, 1)
    assert np.array_equal(fb[..., :3], fb1[..., :3])  # one writer per pixel: the composite is exact


# ----------------------------------------------------------------- known answers from the reference's own code
def test_shade_known_answers(ref_vectors):
    n_ok = 0
    for c in ref_vectors["shade_cases"]:
        mat = np.frombuffer(bytes.fromhex(c["mat"]), orc.MATERIAL_DTYPE)
        ray = np.frombuffer(bytes.fromhex(c["ray"]), orc.RAY_DTYPE)
        light = np.zeros(1, orc.LIGHT_DTYPE)
        light["type"] = c["light_type"]
        light["position"] = c["lpos"]
        light["color"] = c["lcolor"]
        light["normal"] = c["lnormal"]
        light["width"] = c["lwidth"]
        light["height"] = c["lheight"]
        ok, col = orc.shade(mat, ray, c["N"], light, c["sample"])
        assert int(ok) == c["ok"]
        if ok:
            n_ok += 1
            assert col.tobytes().hex() == c["color_hex"], "Shade differs from the reference (material %d, light %d)" % (mat["type"][0], c["light_type"])
    assert n_ok > 30


def test_shade_embree_brdfs_known_answers(ref_vectors):
    """EMBREE_MATERIAL_METAL / VELVET / MATTE (Material.cpp:106-122 -> adapter/embree/EmbreeMaterial.h, the Embree adapter's build of
    Shade): answers of the reference's own code (vendored embree-shaders math).  Embree's rcp/rsqrt are SSE estimates + a Newton step,
    so the restatement is compared to 1e-5 on [0,1] radiance (north_star's float tolerance), not to the bit."""
    n_ok, lit = 0, {3: 0, 4: 0, 5: 0}
    for c in ref_vectors["shade_cases_embree"]:
        mat = np.frombuffer(bytes.fromhex(c["mat"]), orc.MATERIAL_DTYPE)
        ray = np.frombuffer(bytes.fromhex(c["ray"]), orc.RAY_DTYPE)
        light = np.zeros(1, orc.LIGHT_DTYPE)
        light["type"] = c["light_type"]
        light["position"] = c["lpos"]
        light["color"] = c["lcolor"]
        light["normal"] = c["lnormal"]
        light["width"] = c["lwidth"]
        light["height"] = c["lheight"]
        ok, col = orc.shade(mat, ray, c["N"], light, c["sample"])
        assert int(ok) == c["ok"]
        if ok:
            n_ok += 1
            assert np.allclose(col, np.array(c["color"], np.float32), rtol=1e-5, atol=1e-5), (int(mat["type"][0]), col, c["color"])
            if max(c["color"]) > 1e-3:
                lit[int(mat["type"][0])] += 1
    assert n_ok > 60 and min(lit.values()) >= 8, lit


def test_shade_survey_probe():
    """default material, N=(0,0,1), light (0,.1,.5), hit at t=.25 -> (0.5,0.5,0.5)."""
    ray = np.zeros(1, orc.RAY_DTYPE)
    ray["origin"] = (0, 0.1, 0.25)
    ray["direction"] = (0, 0, -1)
    ray["t"] = 0.25
    ray["w"] = 1.0
    ok, c = orc.shade(orc.default_material(), ray, (0, 0, 1), orc.point_light((0, 0.1, 0.5)), (0, 0.1, 0.5))
    assert ok and np.allclose(c, 0.5)


def test_randengine_streams(ref_vectors):
    for rec in ref_vectors["rng"]:
        s = rec["seed"]
        for v in rec["values"]:
            got, s = orc.rng(s)
            assert np.float32(got) == np.float32(v)
        assert s == rec["seed_after"]
    for rec in ref_vectors["lcg"]:
        s = rec["seed"]
        for v in rec["values"]:
            got, s = orc.fastrand_lcg(s)
            assert np.float32(got) == np.float32(v)
        assert s == rec["seed_after"]


def test_generate_normals(ref_vectors):
    v, t = scenes.read_obj(os.path.join(GOLDEN, "bunny.obj"))
    assert v.shape == (2503, 3) and t.shape == (4968, 3)
    assert hashlib.sha256(orc.generate_normals(v, t).tobytes()).hexdigest() == ref_vectors["bunny_normals_sha256"]
    cone = scenes.simple_scene().meshes[0]
    assert orc.generate_normals(cone.verts, cone.tris).tobytes().hex() == ref_vectors["cone_normals_hex"]


def test_add_face_degenerate_filter(ref_vectors):
    rec = ref_vectors["add_faces"]
    kept = scenes.add_faces_1based(np.array(rec["verts"], np.float32), rec["faces1"])
    assert kept.tolist() == rec["kept0"]
    # the cone of SimpleApp.cpp has no degenerate faces, the cube neither
    sc = scenes.simple_scene()
    assert sc.meshes[0].tris.shape == (6, 3) and sc.meshes[1].tris.shape == (12, 3)


def test_raypacket_box_test(ref_vectors):
    """RayPacketIntersection::intersect(update=true) per ray == the oracle's top-level step."""
    for rec in ref_vectors["raypacket"]:
        ray = np.frombuffer(bytes.fromhex(rec["ray"]), orc.RAY_DTYPE).copy()
        ray["t_max"] = np.float32(rec["t_in"])
        nxt, t = orc.toplevel_intersect([rec["lo"]], [rec["hi"]], [0], ray)
        # the packet test has no epsilon... it has: update=true applies tnear > RAY_EPSILON, same as the oracle
        assert int(nxt[0] >= 0) == rec["hit"]
        if rec["hit"]:
            assert np.float32(t[0]).tobytes().hex() == rec["t_out_hex"]


def test_box3d_helpers_and_toplevel_order(ref_vectors):
    sc = scenes.simple_scene()
    order = orc.toplevel_order(sc.inst_lo, sc.inst_hi)
    assert sorted(order.tolist()) == list(range(25))
    one = orc.toplevel_order(sc.inst_lo[:1], sc.inst_hi[:1])
    assert one.tolist() == [0]


def test_ray_constructor_image(ref_vectors):
    """Ray(origin, dir, w, type): direction normalized, t_min = eps, t_max = t = FLT_MAX, id = -1 (Ray.h:106-116)."""
    r = np.frombuffer(bytes.fromhex(ref_vectors["ray_ctor_hex"]), orc.RAY_DTYPE)[0]
    assert np.allclose(r["direction"], (0, 0.6, 0.8)) and r["t_min"] == np.float32(1e-6)
    assert r["t_max"] == np.finfo(np.float32).max and r["t"] == np.finfo(np.float32).max and r["id"] == -1 and r["type"] == 1


def test_area_light_positions(ref_vectors):
    """AreaLight::GetPosition through the oracle's trace path is exercised in test_oracle_consistency; here the LCG only."""
    for rec in ref_vectors["area_light"]:
        s = rec["seed"]
        _, s = orc.fastrand_lcg(s)
        _, s = orc.fastrand_lcg(s)
        assert s == rec["seed_after"]


# ----------------------------------------------------------------- live comparison when the reference build is present
@pytest.mark.skipif(orc.ref() is None, reason="oracle/_ref not built (no /root/reference on this machine)")
def test_live_against_reference_build():
    ref = orc.ref()
    assert ref.ref_sizeof_ray() == 80 and ref.ref_sizeof_material() == 92 and ref.ref_sizeof_box3d() == 32
    rng = np.random.default_rng(5)
    for k in range(200):
        mat = orc.default_material()
        mat["type"] = k % 3
        mat["kd"] = rng.random(3, dtype=np.float32)
        mat["alpha"] = np.float32(1 + 5 * rng.random())
        ray = np.zeros(1, orc.RAY_DTYPE)
        ray["origin"] = rng.random(3, dtype=np.float32)
        d = rng.random(3, dtype=np.float32) - 0.5
        ray["direction"] = d / np.linalg.norm(d)
        ray["t"] = np.float32(rng.random() + 0.1)
        ray["w"] = 1.0
        N = rng.random(3, dtype=np.float32) - 0.5
        N = (N / np.linalg.norm(N)).astype(np.float32)
        lpos = (rng.random(3, dtype=np.float32) * 3).astype(np.float32)
        lcol = rng.random(3, dtype=np.float32)
        c = np.zeros(3, np.float32)
        z = np.zeros(3, np.float32)
        ok_ref = ref.ref_shade(mat.ctypes.data_as(C.c_void_p), ray.ctypes.data_as(C.c_void_p), N.ctypes.data_as(C.c_void_p), C.c_int(0),
                               lpos.ctypes.data_as(C.c_void_p), lcol.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p),
                               C.c_float(0), C.c_float(0), lpos.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p))
        ok, col = orc.shade(mat, ray, N, orc.point_light(lpos, lcol), lpos)
        assert int(ok) == ok_ref
        if ok:
            assert col.tobytes() == c.tobytes()


def test_bounce_math_against_the_libm_the_reference_calls():
    """include/gvt_math.h (the written-out acos / sinf / cosf shared by the oracle and the device code) against the host libm calls
    CosWeightedRandomHemisphereDirection2 makes (EmbreeMeshAdapter.cpp:296-301), over every 16th value RandEngine::rng can return
    (k / 2^24, RandEngine.h:55): theta identical in every case; sin / cos never more than 1 ulp from glibc's (which is not
    correctly rounded: 0.56 ulp) and identical to the correctly rounded (float)sin((double)x)."""
    xi = (np.arange(0, 1 << 24, 16, dtype=np.float64) / float(1 << 24)).astype(np.float32)
    theta = orc.math_probe(2, xi)
    assert (theta.view(np.uint32) == orc.math_probe(18, xi).view(np.uint32)).all()
    phi = (2.0 * 3.1415926535897932384626433832795 * xi.astype(np.float64)).astype(np.float32)
    for arg in (theta, phi):
        for kind, f in ((0, np.sin), (1, np.cos)):
            mine, libm = orc.math_probe(kind, arg), orc.math_probe(kind + 16, arg)
            ulp = np.abs(mine.view(np.int32).astype(np.int64) - libm.view(np.int32).astype(np.int64))
            assert ulp.max() <= 1 and (ulp == 0).mean() > 0.97
            assert (mine.view(np.uint32) == f(arg.astype(np.float64)).astype(np.float32).view(np.uint32)).all()


def test_bounce_direction_known_answers():
    """cos_weighted_dir: unit length, in the hemisphere of n, two draws per call, and a fixed vector for seed 0 (regression pin)."""
    n = np.array([0.0, 0.0, 1.0], np.float32)
    d, s1 = orc.cos_weighted_dir(n, 0)
    _, s_a = orc.rng(0)
    _, s_b = orc.rng(s_a)
    assert s1 == s_b and abs(float(np.linalg.norm(d)) - 1.0) < 1e-6 and d[2] > 0
    acc = np.zeros(3)
    s = 12345
    for _ in range(4000):
        d, s = orc.cos_weighted_dir(n, s)
        assert d[2] >= 0
        acc += d
    assert abs(acc[2] / 4000 - 2.0 / 3.0) < 0.02 and abs(acc[0]) / 4000 < 0.03  # E[cos theta] of a cosine-weighted lobe = 2/3


def test_depth2_frame_does_not_depend_on_rank_count():
    """Carried RNG streams: a depth-2, 4-rays-per-pixel frame of the config-5 stand-in is the same image on 1, 2 and 4 ranks."""
    one = scenes.cathedral_scene(96, 96, samples=2, depth=2, eye=(0.0, 1.5, 13.0), light=(0.0, 2.5, 12.0))
    sc = scenes.split_into_domains(one, 4)
    f1, s1 = oracle_render_domain(sc, [0] * 4, 1, 0)
    f2, s2 = oracle_render_domain(sc, [0, 1, 0, 1], 2, 0)
    f4, s4 = oracle_render_domain(sc, [0, 1, 2, 3], 4, 0)
    assert (f1[..., :3].sum(axis=2) > 0).mean() > 0.2
    assert np.abs(f1 - f2).max() < 1e-6 and np.abs(f1 - f4).max() < 1e-6
    assert np.array_equal(f1[..., 3], f4[..., 3]) and s1.rays_closest == s4.rays_closest and s2.rays_any == s4.rays_any
    assert s1.rays_sent == 0 and s4.rays_sent >= s2.rays_sent > 0


def test_round2_definitions_are_stable():
    """tests/golden/round2_vectors.json pins what the oracle DEFINES where the reference is schedule dependent or calls libm: the
    written-out sin / cos / acos, the camera's stream words, bounce directions, and a depth-2, 4-rays-per-pixel config-5 frame through
    both restated schedulers (image bits, deposit counts, ray counts, rays sent, rounds)."""
    import json

    v = json.load(open(os.path.join(GOLDEN, "round2_vectors.json")))
    xs = np.array(v["math"]["x"], np.uint32).view(np.float32)
    assert orc.math_probe(0, xs).view(np.uint32).tolist() == v["math"]["sin"]
    assert orc.math_probe(1, xs).view(np.uint32).tolist() == v["math"]["cos"]
    assert orc.math_probe(2, np.clip(xs, 0, 0.99999994)).view(np.uint32).tolist() == v["math"]["theta_of_xi"]
    sc = scenes.cathedral_scene(5, 3, samples=2, depth=2)
    r = orc.camera_rays(sc.camera.eye, sc.camera.focus, sc.camera.up, sc.camera.fov, 5, 3, 2, 2, 0.0)
    assert r["rng"].tolist() == v["camera_stream_words"] and len(set(v["camera_stream_words"])) == 60
    for b in v["bounce_dirs"]:
        d, s2 = orc.cos_weighted_dir(np.array(b["n"], np.uint32).view(np.float32), b["seed"])
        assert d.view(np.uint32).tolist() == b["dir"] and s2 == b["seed_after"]
    one = scenes.cathedral_scene(96, 96, samples=2, depth=2, eye=(0.0, 1.5, 13.0), light=(0.0, 2.5, 12.0))
    fb, st = oracle_render(one, 0, nthreads=1)
    rec = v["config5_image_96"]
    assert hashlib.sha256(np.ascontiguousarray(fb[..., :3]).tobytes()).hexdigest() == rec["rgb_sha256"]
    assert hashlib.sha256(np.ascontiguousarray(fb[..., 3]).tobytes()).hexdigest() == rec["alpha_sha256"]
    assert (st.rays_closest, st.rays_any) == (rec["rays_closest"], rec["rays_any"])
    fb, st = oracle_render_domain(scenes.split_into_domains(one, 4), [0, 1, 0, 1], 2, 0, nthreads=1)
    rec = v["config5_domain_96"]
    assert hashlib.sha256(np.ascontiguousarray(fb[..., 3]).tobytes()).hexdigest() == rec["alpha_sha256"]
    assert (st.rays_closest, st.rays_any, st.rays_sent, st.rounds) == (rec["rays_closest"], rec["rays_any"], rec["rays_sent"], rec["rounds"])


def test_known_miss_shortcut_of_the_checker_is_off_by_default_and_image_identical_on_the_soup_tiles():
    """The build's shortcut of shuffleRays (gvt_oracle.c "known misses"; not reference behaviour, an opt-in approximation of the library: a
    re-trace can flip an edge-grazing triangle test, tests/test_gpu_native.py case config5_8): off by default -- every pinning test above runs
    the reference's hop-by-hop rule -- and, switched on, it changes no bit of THIS image while fewer rays are traced and sent."""
    from gravit_amd import scenes
    from tests.helpers import oracle_render_domain

    assert not orc.get_skip_known_misses()
    sc = scenes.soup_domains_scene(200_000, 4, 320, 180)
    owner = [0, 1, 2, 3]
    try:
        a, sa = oracle_render_domain(sc, owner, 4, 0, nthreads=4)
        orc.set_skip_known_misses(True)
        b, sb = oracle_render_domain(sc, owner, 4, 0, nthreads=4)
    finally:
        orc.set_skip_known_misses(False)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and (a[..., 3] > 0).sum() > 1000
    assert sb.rays_closest < sa.rays_closest and sb.rays_sent < sa.rays_sent and sb.rounds < sa.rounds and sb.rays_any == sa.rays_any


def test_known_miss_list_of_a_ray():
    """The list in bytes 68..79 of the Ray image: six 16-bit entries (instance + 1), first free entry, the oldest forgotten when full;
    a ray handed back to an instance on its list is walked through it (origin advanced as TracerBase.h:393 would, twice) and ends."""
    lo = np.array([[0, 0, 0], [0.9, 0, 0]], np.float32)   # two boxes that overlap in x = [0.9, 1.0]
    hi = np.array([[1.0, 1, 1], [2.0, 1, 1]], np.float32)
    order = orc.toplevel_order(lo, hi)
    r = np.zeros(1, orc.RAY_DTYPE)
    r["origin"] = (0.95, 0.5, -2.0); r["direction"] = (0, 0, 1); r["t_max"] = np.float32(3.4e38); r["type"] = 0
    try:
        strict = r.copy()
        assert orc.shuffle_step(lo, hi, order, strict, 0)[0] == 1 and not strict["known"].any()  # reference rule: from 0 on to 1, no list
        orc.set_skip_known_misses(True)
        a = r.copy()
        assert orc.shuffle_step(lo, hi, order, a, 0)[0] == 1 and a["known"][0].tolist() == [1, 0, 0, 0, 0, 0]  # not known yet: goes to 1
        assert np.array_equal(a["origin"], strict["origin"])
        b = a.copy()
        assert orc.shuffle_step(lo, hi, order, b, 1)[0] == -1  # back to 0 (known), on to 1 (known), ... until nothing is ahead: ends
        assert b["known"][0].tolist() == [1, 2, 0, 0, 0, 0] and b["origin"][0, 2] > a["origin"][0, 2]
        c = r.copy()
        c["known"] = [[3, 4, 5, 6, 7, 8]]
        orc.shuffle_step(lo, hi, order, c, 0)
        assert c["known"][0].tolist() == [4, 5, 6, 7, 8, 1]  # full: the oldest entry is forgotten
    finally:
        orc.set_skip_known_misses(False)
