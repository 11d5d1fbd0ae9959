"""Would splitting long thin triangles pay on the config-5 hall (deep tree, boxes of long thin triangles overlap)?  Timing potential only:
triangles whose longest edge exceeds L are cut at that edge's midpoint, recursively (same surface, more and tighter boxes -- but other
triangles and primIDs, so no parity claim); whole frames through the native tracer.   usage (GPU box): python tools/hall_split_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_FLAT
from gravit_amd.scheduler import NativeTracer

capi.init(0)
sc = scenes.cathedral_scene(1024, 1024, samples=2, depth=2, eye=(0.0, 1.5, 13.0), light=(0.0, 2.5, 12.0))
m = sc.meshes[0]
v = m.verts[m.tris.reshape(-1)].reshape(-1, 3, 3).astype(np.float64)

def split(t, L):
    out = []
    while len(t):
        e = np.stack([np.linalg.norm(t[:, 1] - t[:, 0], axis=1), np.linalg.norm(t[:, 2] - t[:, 1], axis=1), np.linalg.norm(t[:, 0] - t[:, 2], axis=1)], 1)
        k = e.argmax(1); big = e.max(1) > L
        out.append(t[~big]); t = t[big]; k = k[big]
        if not len(t): break
        i0, i1, i2 = k, (k + 1) % 3, (k + 2) % 3
        r = np.arange(len(t))
        a, b, c = t[r, i0], t[r, i1], t[r, i2]
        mid = (a + b) * 0.5
        t = np.concatenate([np.stack([a, mid, c], 1), np.stack([mid, b, c], 1)])
    return np.concatenate(out)

edges = np.stack([np.linalg.norm(v[:, 1] - v[:, 0], axis=1), np.linalg.norm(v[:, 2] - v[:, 1], axis=1), np.linalg.norm(v[:, 0] - v[:, 2], axis=1)], 1).max(1)
print("hall: %d triangles, longest edge: median %.3f, p90 %.3f, p99 %.3f, max %.3f" % (len(v), np.median(edges), np.percentile(edges, 90), np.percentile(edges, 99), edges.max()), flush=True)
for L in (1e9, 4.0, 2.0, 1.0, 0.5, 0.25):
    t = split(v, L)
    vv = np.ascontiguousarray(t.reshape(-1, 3).astype(np.float32))
    sc.meshes[0] = scenes.MeshData(vv, np.arange(len(vv), dtype=np.int32).reshape(-1, 3), m.material)
    tr = NativeTracer(sc, NORMALS_FLAT)
    for _ in range(4): tr()
    capi.synchronize(); t0 = time.perf_counter()
    for _ in range(10): tr()
    capi.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    print("max edge %-6s %8d triangles: %.3f ms per frame (%d rays)" % ("-" if L > 1e8 else L, len(t), ms, tr.stats["rays_closest"] + tr.stats["rays_any"]), flush=True)
    tr.close()
