import sys, ctypes, os
sys.path.insert(0, '/root/repo')
import numpy as np
from gravit_amd import capi, scenes
capi.LIB_PATH = '/root/repo/tools/libgvt_hip_stamp.so'
from gravit_amd.adapter import HipMeshAdapter
from oracle import orc
capi.init(0)
for a in sys.argv[1:]:
    if '=' in a:
        k, v = a.split('='); capi.set_option(k, int(v))
sc = scenes.soup_scene(10_000_000)
ad = HipMeshAdapter(sc.meshes[0])
c = sc.camera
rays = orc.camera_rays(c.eye, c.focus, c.up, c.fov, c.width, c.height)
nxt, t = orc.toplevel_intersect(sc.inst_lo, sc.inst_hi, [0], rays)
r = rays[nxt >= 0]
o = r['origin'] if 'primary' in sys.argv else r['origin'] + r['direction'] * (t[nxt >= 0] * np.float32(0.95))[:, None]
# tile order like the frame
W = 1020
idx = np.arange(len(o)).reshape(-1, W); H = idx.shape[0]
ii = idx[:H//8*8, :W//8*8].reshape(H//8, 8, W//8, 8).transpose(0,2,1,3).reshape(-1)
rest = np.setdiff1d(np.arange(len(o)), ii)
ii = np.concatenate([ii, rest])
o, d = o[ii], r['direction'][ii]
lib = capi.load()
lib.gvt_hip_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 24)()
ad.intersect(o, d)
lib.gvt_hip_debug_stamps(buf, 1)
capi.profile(2); capi.stats(True)
ad.intersect(o, d)
st = capi.stats(True); print('kernel ms (stamped build): closest %.4f long %.4f' % (st['ms_closest'], st['ms_long']))
lib.gvt_hip_debug_stamps(buf, 0)
names = ["refill", "inner", "leaf", "retire", "inner_iters", "outer_iters", "waves", "wave_cycles", "drain_cycles", "drain_inner", "drain_outer", "act_at_exh", "max_drain_cyc", "max_drain_inner", "inner_lanes", "leaf_lanes"]
dd = dict(zip(names, list(buf)))
w = dd["waves"]
print(dd)
print("per wave: cycles %.0f refill %.0f inner %.0f leaf %.0f retire %.0f | inner iters %.1f (%.0f each) outer %.1f (leaf %.0f each) | drain cycles %.0f (%.0f%%) drain inner %.1f act at exh %.1f" % (
  dd["wave_cycles"]/w, dd["refill"]/w, dd["inner"]/w, dd["leaf"]/w, dd["retire"]/w, dd["inner_iters"]/w, dd["inner"]/max(1,dd["inner_iters"]), dd["outer_iters"]/w, dd["leaf"]/max(1,dd["outer_iters"]), dd["drain_cycles"]/w, 100*dd["drain_cycles"]/dd["wave_cycles"], dd["drain_inner"]/w, dd["act_at_exh"]/w))
print("lane utilisation: inner loop %.1f%% of 64 lanes per iteration, leaf phase %.1f%% per outer iteration; node steps per ray %.1f, leaf visits per ray %.2f" % (
  100.0 * dd["inner_lanes"] / (64.0 * dd["inner_iters"]), 100.0 * dd["leaf_lanes"] / (64.0 * dd["outer_iters"]), dd["inner_lanes"] / len(o), dd["leaf_lanes"] / len(o)))
b = list(buf)
print("raw", [hex(x) for x in b[16:20]])
t0 = (~b[16]) & 0xFFFFFFFFFFFFFFFF
print("timeline (ticks from the first wave's start): kernel ends %d, last wave exhausted at %d, mean exhaustion %.0f after own start, mean wave life %.0f" % (b[17] - t0, b[18] - t0, b[19] / w, dd["wave_cycles"] / w))
print("wave life: mean %.0f max %d ticks; exhaustion after own start: mean %.0f min %d max %d; mean life / max life = %.2f" % (dd["wave_cycles"] / w, b[20], b[19] / w, (~b[22]) & 0xFFFFFFFFFFFFFFFF, b[21], dd["wave_cycles"] / w / b[20]))
