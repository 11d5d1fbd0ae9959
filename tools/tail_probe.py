"""Closest-hit launch time against ray count on the 10 M-triangle soup (production build): T(n) = a + b n separates the fixed
tail of a launch from its throughput.   usage (GPU box): python tools/tail_probe.py [opt=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi, scenes
from gravit_amd.adapter import HipMeshAdapter
from oracle import orc  # ray generation only (a tool, not the product path)

capi.init(0)
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("="); capi.set_option(k, int(v))
sc = scenes.soup_scene(10_000_000)
ad = HipMeshAdapter(sc.meshes[0])
c = sc.camera
capi.profile(2)
rows = []
rows_any = []
for (w, h) in ((960, 540), (1358, 764), (1920, 1080), (2716, 1528), (3840, 2160)):
    rays = orc.camera_rays(c.eye, c.focus, c.up, c.fov, w, h)
    nxt, t = orc.toplevel_intersect(sc.inst_lo, sc.inst_hi, [0], rays)
    r = rays[nxt >= 0]
    side = int(round(len(r) ** 0.5))
    idx = np.arange(side * side).reshape(side, side)
    s8 = side // 8 * 8
    ii = idx[:s8, :s8].reshape(s8 // 8, 8, s8 // 8, 8).transpose(0, 2, 1, 3).reshape(-1)  # 8x8 tiles like the frame
    o, d = np.ascontiguousarray(r["origin"][ii]), np.ascontiguousarray(r["direction"][ii])
    ad.intersect(o, d)
    best = (1e9, 0)
    for _ in range(5):
        capi.stats(True)
        ad.intersect(o, d)
        st = capi.stats(True)
        best = min(best, (st["ms_closest"], st["ms_long"]))
    rows.append((len(o), best[0], best[1]))
    print("%8d rays: closest %.4f ms + long %.4f ms  (%.1f Mrays/s)" % (len(o), best[0], best[1], len(o) / best[0] / 1e3), flush=True)
    if "any" in sys.argv:  # shadow rays of the benchmark: from the hit point back to the light at the eye
        h = ad.intersect(o, d)
        k = h["prim"] >= 0
        o2 = np.ascontiguousarray(o[k] + d[k] * (h["t"][k] * np.float32(1 - 1e-4))[:, None]); d2 = np.ascontiguousarray(-d[k])
        ad.occluded(o2, d2)
        ba = 1e9
        for _ in range(5):
            capi.stats(True); ad.occluded(o2, d2); ba = min(ba, capi.stats(True)["ms_any"])
        rows_any.append((len(o2), ba))
        print("%8d shadow rays: any hit %.4f ms  (%.1f Mrays/s)" % (len(o2), ba, len(o2) / ba / 1e3), flush=True)
n = np.array([x[0] for x in rows], float); T = np.array([x[1] for x in rows])
b, a = np.polyfit(n, T, 1)
print("fit: T = %.4f ms + %.4f ms per M rays  -> asymptotic %.0f Mrays/s; at 1.04 M rays the fixed part is %.0f %% of the launch" % (a, b * 1e6, 1e-3 / b, 100 * a / (a + b * 1.04e6)))
if rows_any:
    n = np.array([x[0] for x in rows_any], float); T = np.array([x[1] for x in rows_any])
    b, a = np.polyfit(n, T, 1)
    print("any hit fit: T = %.4f ms + %.4f ms per M rays  -> asymptotic %.0f Mrays/s; at 1.0 M rays the fixed part is %.0f %% of the launch" % (a, b * 1e6, 1e-3 / b, 100 * a / (a + b * 1.0e6)))
