"""How long does a per-tile cost map stay useful when the camera moves?  (EXPERIMENTS.md, open candidates: the closest-hit launch's tiles taken
long-first by the PREVIOUS frame's step classes.)  For a camera B that has moved away from camera A, the closest-hit launch of B's frame is timed with its
8x8-pixel tiles (a) in film order, (b) ordered by B's own per-tile maxima (the bound: a perfect predictor), (c) ordered by A's maxima of the same film tiles
(what a renderer would have from its previous frame).  Benchmark soup, 1080p; the motion is an orbit of the eye about the focus point.
   usage (GPU box): python3 tools/order_probe_motion.py [reps=5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi, scenes
from gravit_amd.adapter import HipMeshAdapter
from oracle import orc  # ray generation only (a tool, not the product path)

capi.init(0)
reps = 5
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k == "reps": reps = int(v)
W, H = 1920, 1080
TW, TH = W // 8, H // 8
sc = scenes.soup_scene(10_000_000)
ad = HipMeshAdapter(sc.meshes[0])
c = sc.camera
eye0, foc = np.array(c.eye, np.float64), np.array(c.focus, np.float64)


def frame(deg):
    """rays of the camera after an orbit of `deg` degrees about the focus (y axis), in film-tile order, only those that enter the scene's box;
    their film tile numbers; per-ray node visits"""
    a = np.radians(deg)
    v = eye0 - foc
    eye = foc + np.array([v[0] * np.cos(a) + v[2] * np.sin(a), v[1], -v[0] * np.sin(a) + v[2] * np.cos(a)])
    rays = orc.camera_rays(tuple(eye), tuple(foc), c.up, c.fov, W, H)
    nxt, t = orc.toplevel_intersect(sc.inst_lo, sc.inst_hi, [0], rays)
    ids = rays["id"].astype(np.int64)
    px, py = ids % W, ids // W
    tile = (py // 8) * TW + (px // 8)
    keep = nxt >= 0
    order = np.lexsort(((px % 8), (py % 8), tile))  # tile-major, rows inside a tile
    order = order[keep[order]]
    r = rays[order]
    o = np.ascontiguousarray(r["origin"] + r["direction"] * (t[order] * np.float32(0.95))[:, None]).astype(np.float32)
    d = np.ascontiguousarray(r["direction"])
    vis = ad.visit_stats(o, d)["counts"][:, 0].astype(np.int64)
    return o, d, tile[order], vis


def tile_max(tile, vis):
    m = np.zeros(TW * TH, np.int64)
    np.maximum.at(m, tile, vis)
    return m


def classes(m, like):
    """8 classes by the quantiles of `like`'s non-empty tiles"""
    e = np.percentile(like[like > 0], [12.5, 25, 37.5, 50, 62.5, 75, 87.5])
    return np.searchsorted(e, m, side="right")


def timed(o, d):
    best = 1e9
    for _ in range(reps):
        capi.stats(True); ad.intersect(o, d); st = capi.stats(True); best = min(best, st["ms_closest"] + st.get("ms_long", 0.0))
    return best


capi.profile(2)
oA, dA, tA, vA = frame(0.0)
mA = tile_max(tA, vA)
ad.intersect(oA, dA)
print("camera A: %d rays in %d tiles; per-tile maximum of the node visits: median %d, p90 %d" % (len(oA), (mA > 0).sum(), np.median(mA[mA > 0]), np.percentile(mA[mA > 0], 90)), flush=True)
print("%8s %10s %12s %14s %16s   correlation of the two maps" % ("orbit", "rays", "film order", "own maxima", "A's maxima"), flush=True)
for deg in (0.0, 0.1, 0.25, 0.5, 1.0, 2.0, 5.0, 10.0):
    o, d, t, v = frame(deg)
    m = tile_max(t, v)
    own = classes(m, m)[t]
    prev = classes(mA, mA)[t]
    res = [timed(o, d)]
    for cls in (own, prev):
        p = np.argsort(-cls, kind="stable")  # classes descending, film order inside a class
        res.append(timed(np.ascontiguousarray(o[p]), np.ascontiguousarray(d[p])))
    both = (m > 0) & (mA > 0)
    print("%7.2f° %10d %9.4f ms %11.4f ms %13.4f ms   %.3f" % (deg, len(o), res[0], res[1], res[2], np.corrcoef(m[both], mA[both])[0, 1]), flush=True)
