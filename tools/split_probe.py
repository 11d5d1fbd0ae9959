"""Would spatial splits pay on the benchmark's soup?  The soup's triangles span ~2 cells of its density, so object-partitioning trees
overlap heavily.  Estimate: the same geometry with every triangle cut into 4 / 16 (midpoint subdivision -- tighter boxes, 4x / 16x the
leaves) against the original, visits per primary ray and closest-hit launch time.  2.5 M triangles at the benchmark's
extent-to-spacing ratio.    usage (GPU box): python tools/split_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi, scenes
from gravit_amd.adapter import HipMeshAdapter
from oracle import orc  # ray generation only

def subdivide(v):
    t = v.reshape(-1, 3, 3)
    a, b, c = t[:, 0], t[:, 1], t[:, 2]
    ab, bc, ca = (a + b) * 0.5, (b + c) * 0.5, (c + a) * 0.5
    out = np.stack([np.stack([a, ab, ca], 1), np.stack([ab, b, bc], 1), np.stack([ca, bc, c], 1), np.stack([ab, bc, ca], 1)], 1)
    return np.ascontiguousarray(out.reshape(-1, 3).astype(np.float32))

capi.init(0)
capi.profile(2)
N = 2_500_000
he = 0.005 * (10_000_000 / N) ** (1.0 / 3.0)
v, _ = scenes.triangle_soup(N, 12345, he)
sc = scenes.soup_scene(1000)
c = sc.camera
rays = orc.camera_rays(c.eye, c.focus, c.up, c.fov, 1920, 1080)
nxt, t = orc.toplevel_intersect(np.array([[0, 0, 0]], np.float32) - he, np.array([[1, 1, 1]], np.float32) + he, [0], rays)
r = rays[nxt >= 0]
side = int(len(r) ** 0.5); s8 = side // 8 * 8
ii = np.arange(side * side).reshape(side, side)[:s8, :s8].reshape(s8 // 8, 8, s8 // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
o, d = np.ascontiguousarray(r["origin"][ii]), np.ascontiguousarray(r["direction"][ii])
for level in range(3):
    tris = np.arange(len(v), dtype=np.int32).reshape(-1, 3)
    ad = HipMeshAdapter(scenes.MeshData(v, tris, scenes.default_material()))
    ad.intersect(o, d)
    best = 1e9
    for _ in range(4):
        capi.stats(True); h = ad.intersect(o, d); st = capi.stats(True); best = min(best, st["ms_closest"] + st["ms_long"])
    vs = ad.visit_stats(o[::16], d[::16])
    print("%9d triangles (split level %d): closest+long %.4f ms for %d rays; per ray: %.1f binary inner visits, %.2f leaves, %.2f triangle tests; hits %d" % (
        len(tris), level, best, len(o), vs["inner_per_ray"], vs["leaf_per_ray"], vs["tri_tests_per_ray"], int((h["prim"] >= 0).sum())), flush=True)
    ad.close()
    v = subdivide(v)
