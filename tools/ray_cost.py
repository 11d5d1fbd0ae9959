"""Distribution of per-ray traversal cost (binary-tree visit counts from gvt_hip_visit_stats) for the bench frame's primary rays."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi, scenes
from gravit_amd.adapter import HipMeshAdapter
from oracle import orc
capi.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sc = scenes.soup_scene(N)
ad = HipMeshAdapter(sc.meshes[0])
c = sc.camera
rays = orc.camera_rays(c.eye, c.focus, c.up, c.fov, c.width, c.height)
nxt, t = orc.toplevel_intersect(sc.inst_lo, sc.inst_hi, [0], rays)
r = rays[nxt >= 0]
o = r['origin'] + r['direction'] * (t[nxt >= 0] * np.float32(0.95))[:, None]
s = ad.visit_stats(o, r['direction']); cnt = s.pop('counts')
hits = ad.intersect(o, r['direction'])
hit = hits['prim'] >= 0
inner = cnt[:, 0]
print('rays', len(o), 'hit fraction %.4f' % hit.mean())
for tag, m in (('all', np.ones(len(o), bool)), ('hit', hit), ('miss', ~hit)):
    v = inner[m]
    print(tag, 'n', m.sum(), 'inner visits mean %.1f  p50 %d p90 %d p99 %d p99.9 %d max %d  share of all visits %.3f' % (
        v.mean(), *np.percentile(v, [50, 90, 99, 99.9, 100]), v.sum() / inner.sum()))
