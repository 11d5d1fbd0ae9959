"""The two traversal launches of the benchmark frame on their own (closest hit over the 1.03 M primary rays that reach the soup, any hit
over their shadow rays), 1 warm-up + N timed: the program rocprofv3 wraps for per-kernel counters (tools/pmc.sh).
   usage (GPU box): python3 tools/launch_probe.py [reps=3] [opt=value ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi, scenes
from gravit_amd.adapter import HipMeshAdapter
from oracle import orc  # ray generation only (a tool, not the product path)

capi.init(0)
reps = 3
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
o2 = np.ascontiguousarray(o[k] + d[k] * (h["t"][k] * np.float32(1 - 1e-4))[:, None]); d2 = np.ascontiguousarray(-d[k])
ad.occluded(o2, d2)
capi.profile(2)
bc, ba = 1e9, 1e9
for _ in range(reps):
    capi.stats(True); ad.intersect(o, d); st = capi.stats(True); bc = min(bc, st["ms_closest"])
    capi.stats(True); ad.occluded(o2, d2); ba = min(ba, capi.stats(True)["ms_any"])
print("closest %d rays %.4f ms; any hit %d rays %.4f ms" % (len(o), bc, len(o2), ba), flush=True)
