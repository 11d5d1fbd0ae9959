"""Phase breakdown and step counts of k_traceq (four lanes per ray) from s_memtime stamps, with a private -DGVT_STAMP=1 build.
   usage: python tools/stamps.py --build   (here; cross-compiles tools/libgvt_hip_stamp.so)
          python tools/stamps_quad.py [closest|any] [opt=value ...]   (GPU box): one launch over the benchmark frame's 1.03 M rays"""
import sys, ctypes, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi, scenes
capi.LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgvt_hip_stamp.so")
from gravit_amd.adapter import HipMeshAdapter
from oracle import orc  # ray generation only (a tool, not the product path)
capi.init(0)
kind = "any" if "any" in sys.argv else "closest"
for a in sys.argv[1:]:
    if "=" in a:
        k, v = a.split("="); capi.set_option(k, int(v))
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
if kind == "any":
    k = h["prim"] >= 0
    o = np.ascontiguousarray(o[k] + d[k] * (h["t"][k] * np.float32(1 - 1e-4))[:, None]); d = np.ascontiguousarray(-d[k])
run = (lambda: ad.occluded(o, d)) if kind == "any" else (lambda: ad.intersect(o, d))
run()
buf = (ctypes.c_ulonglong * 24)()
lib = capi.load()
lib.gvt_hip_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
lib.gvt_hip_debug_stamps(buf, 1)
run(); capi.synchronize()
lib.gvt_hip_debug_stamps(buf, 0)
v = list(buf); n = len(o); w = max(1, v[6])
print("%s, %d rays, %d waves: cycles per wave %.0f = refill %.0f + inner %.0f + leaf %.0f + retire %.0f" % (kind, n, w, v[7] / w, v[0] / w, v[1] / w, v[2] / w, v[3] / w))
print("inner iterations per wave %.1f (%.0f cycles each, %.2f of 16 quads busy); node steps per ray %.1f" % (v[4] / w, v[1] / max(1, v[4]), v[14] / max(1, v[4]), v[14] / n))
print("leaf phases per wave %.1f of %.1f outer iterations (%.0f cycles each, %.2f quads at a leaf); leaf steps per ray %.2f, triangle tests per ray %.2f" % (
    v[9] / w, v[5] / w, v[2] / max(1, v[9]), v[15] / max(1, v[9]), v[15] / n, v[10] / n))
