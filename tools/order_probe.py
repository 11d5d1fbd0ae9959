"""Does the ORDER in which a launch takes its 64-ray tiles shorten its drain?  A persistent-wave launch ends with the dependent chain of
the last rays started (EXPERIMENTS.md: closest T = 0.18 ms + 0.31 ms per M rays); if the tiles that hold the longest rays are started
first, the rays left for the end are short ones.  This probe permutes the benchmark frame's ray lists tile by tile -- by the tile's
maximum node-visit count (from the diagnostic kernel: a perfect predictor, the upper bound of what a previous frame or the primary
ray's own count can give) -- and times the two traversal launches for each order.
   usage (GPU box): python3 tools/order_probe.py [reps=5]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi, scenes
from gravit_amd.adapter import HipMeshAdapter
from oracle import orc  # ray generation only (a tool, not the product path)

capi.init(0)
reps = 5
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("=")
        if k == "reps": reps = int(v)
        else: capi.set_option(k, int(v))
sc = scenes.soup_scene(10_000_000)
ad = HipMeshAdapter(sc.meshes[0])
c = sc.camera
rays = orc.camera_rays(c.eye, c.focus, c.up, c.fov, 1920, 1080)
nxt, t = orc.toplevel_intersect(sc.inst_lo, sc.inst_hi, [0], rays)
r = rays[nxt >= 0]
side = int(round(len(r) ** 0.5)); idx = np.arange(side * side).reshape(side, side); s8 = side // 8 * 8
ii = idx[:s8, :s8].reshape(s8 // 8, 8, s8 // 8, 8).transpose(0, 2, 1, 3).reshape(-1)
o, d = np.ascontiguousarray(r["origin"][ii]), np.ascontiguousarray(r["direction"][ii])
h = ad.intersect(o, d)
k = h["prim"] >= 0
# shadow rays in the primaries' slots (tiles stay tiles); a primary that missed leaves a copy of its neighbour (a handful of rays)
o2 = np.ascontiguousarray(o + d * (h["t"] * np.float32(1 - 1e-4))[:, None]); d2 = np.ascontiguousarray(-d)
o2[~k] = o2[k][0]; d2[~k] = d2[k][0]
vp = ad.visit_stats(o, d)["counts"][:, 0].astype(np.int64)   # inner visits of the binary tree per primary ray
vs = ad.visit_stats(o2, d2)["counts"][:, 0].astype(np.int64)  # ... per shadow ray
nt = len(o) // 64
print("rays %d, tiles %d; primary visits mean %.1f p99 %d max %d; shadow visits mean %.1f p99 %d max %d; corr(primary, shadow) = %.3f"
      % (len(o), nt, vp.mean(), np.percentile(vp, 99), vp.max(), vs.mean(), np.percentile(vs, 99), vs.max(), np.corrcoef(vp, vs)[0, 1]), flush=True)


def orders(cost):
    tmax = cost[:nt * 64].reshape(nt, 64).max(axis=1)
    out = {"natural": np.arange(nt)}
    for q in (95, 80):
        thr = np.percentile(tmax, q)
        out["tiles above p%d first" % q] = np.concatenate([np.nonzero(tmax > thr)[0], np.nonzero(tmax <= thr)[0]])
    qs = np.percentile(tmax, [75, 50, 25])
    b = np.zeros(nt, np.int64)
    for x in qs: b += (tmax <= x)
    out["4 buckets, each in list order"] = np.argsort(b, kind="stable")
    e = np.percentile(tmax, [87.5, 75, 62.5, 50, 37.5, 25, 12.5])
    b = np.zeros(nt, np.int64)
    for x in e: b += (tmax <= x)
    out["8 buckets, each in list order"] = np.argsort(b, kind="stable")
    out["descending by tile maximum"] = np.argsort(-tmax, kind="stable")
    # the launch hands every wave a STATIC first chunk of about two tiles and the rest dynamically: descending order gives the first waves two long tiles each.
    # Interleaved: the static chunks hold one long and one short tile each (descending order folded onto itself), the dynamic part the middle
    d_ = np.argsort(-tmax, kind="stable")
    n_static = min(nt, 2 * 5120) // 2 * 2
    head, tail = d_[:n_static // 2], d_[nt - n_static // 2:][::-1]
    out["static chunks long + short, then the middle descending"] = np.concatenate([np.stack([head, tail], 1).reshape(-1), d_[n_static // 2: nt - n_static // 2]])
    e8 = np.percentile(tmax, [87.5, 75, 62.5, 50, 37.5, 25, 12.5])
    b8 = np.zeros(nt, np.int64)
    for x in e8: b8 += (tmax <= x)
    rr = np.argsort(b8, kind="stable")  # 8 buckets; then dealt out round-robin over 5120 waves' static pairs: tile k of the order goes to pair (k mod 5120)
    if nt >= 2 * 5120:
        first = rr[:2 * 5120].reshape(2, 5120).T.reshape(-1)
        out["8 buckets, the static pairs dealt round-robin"] = np.concatenate([first, rr[2 * 5120:]])
    out["ascending (worst case)"] = np.argsort(tmax, kind="stable")
    return out


def expand(tile_order):
    return (tile_order[:, None] * 64 + np.arange(64)[None, :]).reshape(-1)


capi.profile(2)
ad.intersect(o, d); ad.occluded(o2, d2)
print("closest hit, tiles ordered by the primaries' own counts:", flush=True)
for name, to in orders(vp).items():
    p = expand(to); oo, dd = np.ascontiguousarray(o[p]), np.ascontiguousarray(d[p])
    best = 1e9
    for _ in range(reps):
        capi.stats(True); ad.intersect(oo, dd); st = capi.stats(True); best = min(best, st["ms_closest"] + st.get("ms_long", 0.0))
    print("  %-56s %.4f ms" % (name, best), flush=True)
for label, cost in (("the shadow rays' own counts (upper bound)", vs), ("their primaries' counts (known inside the frame)", vp)):
    print("any hit, tiles ordered by %s:" % label, flush=True)
    for name, to in orders(cost).items():
        p = expand(to); oo, dd = np.ascontiguousarray(o2[p]), np.ascontiguousarray(d2[p])
        best = 1e9
        for _ in range(reps):
            capi.stats(True); ad.occluded(oo, dd); best = min(best, capi.stats(True)["ms_any"])
        print("  %-56s %.4f ms" % (name, best), flush=True)
