#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.  Run in the build container only
(it reads /root/reference and the reference build oracle/_ref/libgvtref.so):

    make -C oracle && python tests/golden/make_fixtures.py

Fixtures are DATA (inputs and expected outputs), never reference source text:
  ref_simple.ppm, ref_bunny.ppm  the reference's own golden images (Test/CTESTtest/data/), copied as they are
  bunny.obj                      the reference's data/geom/bunny.obj (input of the bunny golden)
  bun_zipper.npz                 vertices/faces of data/geom/bunny/reconstruction/bun_zipper.ply (config 2 input)
  ref_vectors.json               known answers produced by the REFERENCE's own code (Shade, Light, generateNormals,
                                 RayPacketIntersection, RandEngine, Mesh::addFace, struct sizes) through oracle/ref_shim.cpp
  oracle_vectors.npz             outputs of the pinned CPU oracle on seeded inputs (per-ray hits on bunny.obj, framebuffer
                                 hashes of the golden scenes in flat and smooth mode) for the GPU parity tests
  round2_vectors.json            (python tests/golden/make_fixtures.py --round2; needs no reference tree) regression pins of what
                                 the oracle DEFINES where the reference is schedule dependent: gvt_math known answers, camera stream
                                 words, bounce directions, and the hash / ray counts of a depth-2, 4-rays-per-pixel config-5 frame
"""
import ctypes as C
import hashlib
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from gravit_amd import scenes  # noqa: E402
from oracle import orc  # noqa: E402


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def f3(*v):
    return np.array(v, np.float32)


def round2():
    """Pins of the oracle's own definitions (no reference involved): written once, checked by tests/test_oracle_pinning.py."""
    from tests.helpers import oracle_render, oracle_render_domain

    out = {}
    xs = np.array([0.0, 1e-7, 0.25, 0.5, 0.999999, 1.0, 3.14159274, 6.28318548, 2.5, 5.75], np.float32)
    out["math"] = {"x": xs.view(np.uint32).tolist(), "sin": orc.math_probe(0, xs).view(np.uint32).tolist(),
                   "cos": orc.math_probe(1, xs).view(np.uint32).tolist(),
                   "theta_of_xi": orc.math_probe(2, np.clip(xs, 0, 0.99999994)).view(np.uint32).tolist()}
    sc = scenes.cathedral_scene(5, 3, samples=2, depth=2)
    r = orc.camera_rays(sc.camera.eye, sc.camera.focus, sc.camera.up, sc.camera.fov, 5, 3, 2, 2, 0.0)
    out["camera_stream_words"] = r["rng"].tolist()
    dirs = []
    for k, n in enumerate(([0, 0, 1], [0.6, 0.0, 0.8], [-0.57735026, 0.57735026, 0.57735026])):
        d, s2 = orc.cos_weighted_dir(np.array(n, np.float32), 1000 + k)
        dirs.append({"n": np.array(n, np.float32).view(np.uint32).tolist(), "seed": 1000 + k, "dir": d.view(np.uint32).tolist(), "seed_after": int(s2)})
    out["bounce_dirs"] = dirs
    one = scenes.cathedral_scene(96, 96, samples=2, depth=2, eye=(0.0, 1.5, 13.0), light=(0.0, 2.5, 12.0))
    fb, st = oracle_render(one, 0, nthreads=1)
    out["config5_image_96"] = {"rgb_sha256": hashlib.sha256(np.ascontiguousarray(fb[..., :3]).tobytes()).hexdigest(),
                               "alpha_sha256": hashlib.sha256(np.ascontiguousarray(fb[..., 3]).tobytes()).hexdigest(),
                               "rays_closest": int(st.rays_closest), "rays_any": int(st.rays_any)}
    dom = scenes.split_into_domains(one, 4)
    fb, st = oracle_render_domain(dom, [0, 1, 0, 1], 2, 0, nthreads=1)
    out["config5_domain_96"] = {"alpha_sha256": hashlib.sha256(np.ascontiguousarray(fb[..., 3]).tobytes()).hexdigest(),
                                "rays_closest": int(st.rays_closest), "rays_any": int(st.rays_any), "rays_sent": int(st.rays_sent), "rounds": int(st.rounds)}
    json.dump(out, open(os.path.join(HERE, "round2_vectors.json"), "w"), indent=1, sort_keys=True)
    print("wrote round2_vectors.json")


def main():
    if "--round2" in sys.argv:
        return round2()
    assert os.path.isdir(REF), "needs /root/reference"
    ref = orc.ref()
    assert ref is not None, "build oracle/_ref first (make -C oracle)"
    shutil.copyfile(os.path.join(REF, "Test/CTESTtest/data/simple.ppm"), os.path.join(HERE, "ref_simple.ppm"))
    shutil.copyfile(os.path.join(REF, "Test/CTESTtest/data/bunny.ppm"), os.path.join(HERE, "ref_bunny.ppm"))
    shutil.copyfile(os.path.join(REF, "data/geom/bunny.obj"), os.path.join(HERE, "bunny.obj"))
    v, t = scenes.read_ply(os.path.join(REF, "data/geom/bunny/reconstruction/bun_zipper.ply"))
    np.savez_compressed(os.path.join(HERE, "bun_zipper.npz"), verts=v, tris=t)

    out = {"sizes": {"ray": ref.ref_sizeof_ray(), "material": ref.ref_sizeof_material(), "box3d": ref.ref_sizeof_box3d()},
           "ray_epsilon": float(ref.ref_ray_epsilon())}

    # ---- default material image
    mat = np.zeros(1, orc.MATERIAL_DTYPE)
    ref.ref_default_material(p(mat))
    out["default_material_hex"] = mat.tobytes().hex()

    # ---- Shade known answers: materials x lights x geometry
    rng = np.random.default_rng(1234)
    cases = []
    for k in range(96):
        m = mat.copy()
        m["type"] = k % 3
        m["kd"] = rng.random(3, dtype=np.float32)
        m["ks"] = rng.random(3, dtype=np.float32)
        m["alpha"] = np.float32(1 + 7 * rng.random())
        ray = np.zeros(1, orc.RAY_DTYPE)
        ray["origin"] = rng.random(3, dtype=np.float32) * 2 - 1
        d = rng.random(3, dtype=np.float32) * 2 - 1
        ray["direction"] = d / np.linalg.norm(d)
        ray["t"] = np.float32(0.2 + rng.random())
        ray["w"] = np.float32(0.25 + 0.75 * rng.random())
        N = rng.random(3, dtype=np.float32) * 2 - 1
        N = (N / np.linalg.norm(N)).astype(np.float32)
        lt = (k // 3) % 3
        lpos = (rng.random(3, dtype=np.float32) * 4 - 2).astype(np.float32)
        lcol = rng.random(3, dtype=np.float32)
        lnorm = f3(0, -1, 0) if k % 2 else f3(0.3, 0.8, 0.1)
        lw, lh = np.float32(0.5), np.float32(0.25)
        sample = lpos + f3(0.01, 0.02, -0.03) if lt == 1 else lpos
        c = np.zeros(3, np.float32)
        ok = ref.ref_shade(p(m), p(ray), p(N), C.c_int(lt), p(lpos), p(lcol), p(lnorm), C.c_float(lw), C.c_float(lh), p(sample), p(c))
        cases.append({"mat": m.tobytes().hex(), "ray": ray.tobytes().hex(), "N": N.tolist(), "light_type": lt, "lpos": lpos.tolist(),
                      "lcolor": lcol.tolist(), "lnormal": lnorm.tolist(), "lwidth": float(lw), "lheight": float(lh),
                      "sample": sample.tolist(), "ok": int(ok), "color_hex": c.tobytes().hex()})
    # SURVEY probe: default material, N=(0,0,1), light (0,.1,.5), hit at t=.25 -> (0.5,0.5,0.5)
    out["shade_cases"] = cases

    # ---- Shade, Embree-tutorial BRDFs (EMBREE_MATERIAL_METAL / VELVET / MATTE, Material.cpp:106-122 under GVT_RENDER_ADAPTER_EMBREE,
    #      the Embree adapter's build): float answers -- Embree's rcp/rsqrt are SSE estimates, so the restatement matches to ~1e-6
    rng = np.random.default_rng(4321)
    ecases = []
    for k in range(120):
        m = mat.copy()
        m["type"] = 3 + k % 3
        m["kd"] = rng.random(3, dtype=np.float32)
        m["ks"] = rng.random(3, dtype=np.float32)
        m["eta"] = (0.2 + 2.5 * rng.random(3)).astype(np.float32)
        m["k"] = (1.0 + 5.0 * rng.random(3)).astype(np.float32)
        m["roughness"] = np.float32(0.02 + 0.5 * rng.random())
        m["hsc"] = rng.random(3, dtype=np.float32)
        m["backScattering"] = np.float32(0.2 + 1.5 * rng.random())
        m["hsFallOff"] = np.float32(1 + 9 * rng.random())
        ray = np.zeros(1, orc.RAY_DTYPE)
        ray["origin"] = rng.random(3, dtype=np.float32) * 2 - 1
        N = rng.random(3, dtype=np.float32) * 2 - 1
        N = (N / np.linalg.norm(N)).astype(np.float32)
        d = rng.random(3, dtype=np.float32) * 2 - 1
        if k % 4:  # mostly front-facing geometry so that the lobes are exercised
            d = d - N * (np.dot(d, N) + abs(np.dot(d, N)) + 0.2)
        ray["direction"] = d / np.linalg.norm(d)
        ray["t"] = np.float32(0.2 + rng.random())
        ray["w"] = np.float32(0.25 + 0.75 * rng.random())
        hit = ray["origin"][0] + ray["direction"][0] * ray["t"][0]
        lt = (k // 3) % 3
        lpos = (hit + N * np.float32(0.5 + rng.random()) + (rng.random(3, dtype=np.float32) - 0.5)).astype(np.float32) if k % 5 else \
            (rng.random(3, dtype=np.float32) * 4 - 2).astype(np.float32)
        lcol = rng.random(3, dtype=np.float32)
        lnorm = f3(0, -1, 0)
        lw, lh = np.float32(0.5), np.float32(0.25)
        sample = lpos + f3(0.01, 0.02, -0.03) if lt == 1 else lpos
        c = np.zeros(3, np.float32)
        ok = ref.ref_shade(p(m), p(ray), p(N), C.c_int(lt), p(lpos), p(lcol), p(lnorm), C.c_float(lw), C.c_float(lh), p(sample), p(c))
        ecases.append({"mat": m.tobytes().hex(), "ray": ray.tobytes().hex(), "N": N.tolist(), "light_type": lt, "lpos": lpos.tolist(),
                       "lcolor": lcol.tolist(), "lnormal": lnorm.tolist(), "lwidth": float(lw), "lheight": float(lh),
                       "sample": sample.tolist(), "ok": int(ok), "color": [float(x) for x in c]})
    out["shade_cases_embree"] = ecases

    # ---- area light sample positions (LCG stream)
    al = []
    for seed in (0, 1, 12345, 0xDEADBEEF):
        s = C.c_uint32(seed)
        o = np.zeros(3, np.float32)
        ref.ref_area_light_position(p(f3(0.1, 2.0, -0.3)), p(f3(1, 1, 1)), p(f3(0.3, 0.8, 0.1)), C.c_float(0.5), C.c_float(0.25), C.byref(s), p(o))
        al.append({"seed": seed, "pos_hex": o.tobytes().hex(), "seed_after": s.value})
    out["area_light"] = al

    # ---- RandEngine streams
    rs = []
    for seed in (0, 1, 42, 0xFFFFFFFF):
        s = C.c_uint32(seed)
        vals = []
        for _ in range(8):
            vals.append(float(ref.ref_rng(C.byref(s))))
        rs.append({"seed": seed, "values": vals, "seed_after": s.value})
    out["rng"] = rs
    ls = []
    for seed in (0, 7, 99999):
        s = C.c_uint32(seed)
        vals = [float(ref.ref_fastrand_lcg(C.byref(s), C.c_float(0), C.c_float(1))) for _ in range(8)]
        ls.append({"seed": seed, "values": vals, "seed_after": s.value})
    out["lcg"] = ls

    # ---- generateNormals on bunny.obj (hash) and on a tiny mesh (values)
    bv, bt = scenes.read_obj(os.path.join(HERE, "bunny.obj"))
    nrm = np.zeros_like(bv)
    ref.ref_generate_normals(p(bv), C.c_int(len(bv)), p(bt), C.c_int(len(bt)), p(nrm))
    out["bunny_normals_sha256"] = hashlib.sha256(nrm.tobytes()).hexdigest()
    sc = scenes.simple_scene()
    cone = sc.meshes[0]
    cn = np.zeros_like(cone.verts)
    ref.ref_generate_normals(p(cone.verts), C.c_int(len(cone.verts)), p(cone.tris), C.c_int(len(cone.tris)), p(cn))
    out["cone_normals_hex"] = cn.tobytes().hex()

    # ---- Mesh::addFace degenerate filter
    tv = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 0, 0], [0, 0, 1]], np.float32)
    tf = np.array([[1, 2, 3], [2, 4, 5], [1, 3, 5], [2, 2, 3]], np.int32)  # 2nd and 4th have coincident vertices
    tout = np.zeros((8, 3), np.int32)
    k = ref.ref_add_faces(p(tv), C.c_int(len(tv)), p(tf), C.c_int(len(tf)), p(tout))
    out["add_faces"] = {"verts": tv.tolist(), "faces1": tf.tolist(), "kept0": tout[:k].tolist()}

    # ---- RayPacketIntersection<1>::intersect(update=true)
    rp = []
    for k in range(64):
        ray = np.zeros(1, orc.RAY_DTYPE)
        ray["origin"] = rng.random(3, dtype=np.float32) * 4 - 2
        d = rng.random(3, dtype=np.float32) * 2 - 1
        if k % 8 == 0:
            d[k % 3] = 0.0  # axis-parallel component -> inf reciprocal
        ray["direction"] = d / np.linalg.norm(d)
        lo = (rng.random(3, dtype=np.float32) - 0.8).astype(np.float32)
        hi = (lo + rng.random(3, dtype=np.float32)).astype(np.float32)
        tin = np.float32(3.0 if k % 2 else np.finfo(np.float32).max)
        t = C.c_float(tin)
        hit = ref.ref_raypacket_intersect(p(ray), p(lo), p(hi), C.byref(t))
        rp.append({"ray": ray.tobytes().hex(), "lo": lo.tolist(), "hi": hi.tolist(), "t_in": float(tin), "hit": int(hit),
                   "t_out_hex": np.float32(t.value).tobytes().hex()})
    out["raypacket"] = rp

    # ---- Box3D helpers used by the top-level BVH build
    bx = []
    for k in range(8):
        lo = (rng.random(3, dtype=np.float32) - 0.5).astype(np.float32)
        hi = (lo + rng.random(3, dtype=np.float32)).astype(np.float32)
        bx.append({"lo": lo.tolist(), "hi": hi.tolist(), "area_hex": np.float32(ref.ref_box_surface_area(p(lo), p(hi))).tobytes().hex(),
                   "wide": int(ref.ref_box_wide_dir(p(lo), p(hi)))})
    out["box3d"] = bx

    # ---- Ray constructor image
    ro = np.zeros(80, np.uint8)
    ref.ref_ray_ctor(p(f3(1, 2, 3)), p(f3(0, 3, 4)), C.c_float(0.5), C.c_int(1), p(ro))
    out["ray_ctor_hex"] = ro.tobytes().hex()

    json.dump(out, open(os.path.join(HERE, "ref_vectors.json"), "w"), indent=1)

    # ================= oracle-produced vectors (the oracle is pinned on the goldens above) =================
    om = orc.Mesh(bv, bt)
    r2 = np.random.default_rng(2024)
    n = 4096
    lo, hi = bv.min(0), bv.max(0)
    org = (lo + (hi - lo) * r2.random((n, 3))).astype(np.float32)
    org[:, 2] = hi[2] + 0.2
    tgt = (lo + (hi - lo) * r2.random((n, 3))).astype(np.float32)
    dirs = tgt - org
    dirs = (dirs / np.linalg.norm(dirs, axis=1, keepdims=True)).astype(np.float32)
    hits = om.intersect(org, dirs)
    occ = om.occluded(org, dirs)
    fbh = {}
    for name, scn in (("simple", scenes.simple_scene()), ("bunny", scenes.bunny_scene())):
        meshes = [orc.Mesh(m.verts, m.tris, mesh_mat=m.material) for m in scn.meshes]
        cam = scn.camera
        rays = orc.camera_rays(cam.eye, cam.focus, cam.up, cam.fov, cam.width, cam.height, cam.samples, cam.depth, cam.jitter)
        for mode in (0, 1):
            fb, st = orc.render_image([meshes[i] for i in scn.inst_mesh], scn.m, scn.minv, scn.normi, scn.inst_lo, scn.inst_hi,
                                      scn.lights, rays, cam.width, cam.height, mode, 8)
            fbh["%s_mode%d" % (name, mode)] = {"rgb_sha256": hashlib.sha256(np.ascontiguousarray(fb[..., :3]).tobytes()).hexdigest(),
                                               "adapter_calls": int(st.adapter_calls), "rays_closest": int(st.rays_closest),
                                               "rays_any": int(st.rays_any)}
    np.savez_compressed(os.path.join(HERE, "oracle_vectors.npz"), org=org, dirs=dirs, hits=hits, occluded=occ,
                        fb_hashes=json.dumps(fbh))
    print("fixtures written to", HERE)


if __name__ == "__main__":
    main()
