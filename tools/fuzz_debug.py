"""Re-runs ONE seed of tests/test_gpu_parity.py::test_random_meshes_and_rays_against_the_oracle and prints the rays whose hits differ: the device's hit, the
checker's through its own BVH and the checker's brute force over all triangles.   python tools/fuzz_debug.py <seed> [opt=value ...]   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi, scenes
from gravit_amd.adapter import HipMeshAdapter
from oracle import orc

seed = int(sys.argv[1])
capi.init(0)
for a in sys.argv[2:]:
    k, v = a.split("="); capi.set_option(k, int(v))
rng = np.random.default_rng(1000 + seed)
n_v = int(rng.integers(3, 2000)); n_t = int(rng.integers(1, 3000)); scale = 10.0 ** rng.integers(-3, 3)
v = rng.normal(size=(n_v, 3)) * scale
if seed % 3 == 1:
    v[rng.integers(0, n_v, max(1, n_v // 50))] *= 50.0
    v[rng.integers(0, n_v, max(1, n_v // 50))] = v[0] + rng.normal(size=(max(1, n_v // 50), 3)) * scale * 1e-4
if seed % 4 == 2:
    v[:, int(rng.integers(0, 3))] = np.round(v[:, 0] / scale) * scale
v = v.astype(np.float32)
t = rng.integers(0, n_v, (n_t, 3)).astype(np.int32)
if n_t > 4:
    t[rng.integers(0, n_t, n_t // 8 + 1)] = t[rng.integers(0, n_t, n_t // 8 + 1)]
mesh = scenes.MeshData(v, t)
ad, om = HipMeshAdapter(mesh), orc.Mesh(v, t)
lo, hi = v[t.reshape(-1)].min(axis=0), v[t.reshape(-1)].max(axis=0)
ext = np.maximum(hi - lo, 1e-6 * scale)
n = int(rng.integers(1, 5000))
org = (lo - 0.5 * ext + 2.0 * ext * rng.random((n, 3))).astype(np.float32)
tgt = v[t[rng.integers(0, n_t, n)]].astype(np.float64)
w = rng.dirichlet([1.0, 1.0, 1.0], n)
kind = rng.integers(0, 5, n)
w[kind == 1] = [1.0, 0.0, 0.0]
w[kind == 2, 2] = 0.0
w[kind == 2] /= np.maximum(w[kind == 2].sum(axis=1, keepdims=True), 1e-9)
p = (tgt * w[:, :, None]).sum(axis=1)
d = p - org
ax = kind == 3
d[ax] = np.eye(3)[rng.integers(0, 3, ax.sum())] * rng.choice([-1.0, 1.0], (ax.sum(), 1))
inpl = kind == 4
d[inpl] = (tgt[inpl, 1] - tgt[inpl, 0]) + 1e-30
nrm = np.linalg.norm(d, axis=1, keepdims=True)
d = np.where(nrm > 0, d / np.maximum(nrm, 1e-300), [0.0, 0.0, 1.0]).astype(np.float32)
g, c, b = ad.intersect(org, d), om.intersect(org, d), om.intersect(org, d, use_bvh=False)
print("seed %d: %d vertices, %d triangles, scale %g, %d rays; info %s" % (seed, n_v, n_t, scale, n, ad.info()))
bad = np.nonzero((g["prim"] != c["prim"]) | (g["t"].view(np.uint32) != c["t"].view(np.uint32)))[0]
print("device vs checker BVH: %d rays differ; checker BVH vs checker brute force: %d differ; device vs brute force: %d differ" % (
    len(bad), int(((c["prim"] != b["prim"]) | (c["t"].view(np.uint32) != b["t"].view(np.uint32))).sum()), int(((g["prim"] != b["prim"]) | (g["t"].view(np.uint32) != b["t"].view(np.uint32))).sum())))
for i in bad[:10]:
    print(" ray %d kind %d org %s dir %s\n   device %s\n   checker %s\n   brute   %s" % (i, kind[i], org[i], d[i], g[i], c[i], b[i]))
    for name, h in (("device", g[i]), ("checker", c[i])):
        if h["prim"] >= 0:
            tv = v[t[h["prim"]]]
            print("   %s's triangle %d: %s area %.3g" % (name, h["prim"], tv.tolist(), 0.5 * np.linalg.norm(np.cross(tv[1] - tv[0], tv[2] - tv[0]))))
