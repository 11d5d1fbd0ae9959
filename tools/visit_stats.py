import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi, scenes
from gravit_amd.adapter import HipMeshAdapter
from oracle import orc
capi.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
sc = scenes.soup_scene(N)
ad = HipMeshAdapter(sc.meshes[0])
print(ad.info(), flush=True)
c = sc.camera
rays = orc.camera_rays(c.eye, c.focus, c.up, c.fov, c.width, c.height)
nxt, t = orc.toplevel_intersect(sc.inst_lo, sc.inst_hi, [0], rays)
r = rays[nxt >= 0]
o = r['origin'] + r['direction'] * (t[nxt >= 0] * np.float32(0.95))[:, None]
def show(tag, o, d):
    s = ad.visit_stats(o, d); cnt = s.pop('counts')
    st = cnt[:,0]+cnt[:,1]
    print(tag, {k: round(v,2) for k,v in s.items()}, 'steps pct50/90/99/max', np.percentile(st,[50,90,99,100]), flush=True)
show('rows', o, r['direction'])
# 8x8 tiles
W = 1020
idx = np.arange(len(o)).reshape(-1, W)
H = idx.shape[0]
til = idx[:H//8*8].reshape(H//8, 8, W//8*8//8 if False else -1)
ii = idx[:H//8*8, :W//8*8].reshape(H//8, 8, W//8, 8).transpose(0,2,1,3).reshape(-1)
show('tiles8x8', o[ii], r['direction'][ii])
perm = np.random.default_rng(0).permutation(len(o))
show('shuffled', o[perm], r['direction'][perm])
