import sys, time, itertools, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gravit_amd import capi, scenes
from gravit_amd.scheduler import ImageTracer
capi.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sc = scenes.soup_scene(N)
tr = ImageTracer(sc, 0)
def run(opts, frames=4):
    for k, v in opts.items(): capi.set_option(k, v)
    tr(); capi.synchronize()
    capi.stats_reset(); capi.profile(True)
    t = time.perf_counter()
    for _ in range(frames): tr()
    capi.synchronize(); dt = (time.perf_counter() - t) / frames
    st = capi.stats(); capi.profile(False)
    return dt * 1e3, st['ms_closest'] / frames, st['ms_any'] / frames, (st['rays_closest'] + st['rays_any']) / frames / dt / 1e6
configs = []
for bb in (3, 4, 5, 6):
    for (r, i) in ((8, 20), (16, 32), (24, 40), (32, 48)):
        configs.append(dict(trav_kernel=1, blocks_per_cu=bb, refill_min=r, inner_min=i, share=1))
res = {}
for rnd in range(2):
    for c in configs:
        key = json.dumps(c)
        res.setdefault(key, []).append(run(c))
for k, v in res.items():
    a = np.array(v)
    print(k, 'frame %.3f ms closest %.3f any %.3f  Mrays/s %.0f' % (tuple(a.min(axis=0)[:3]) + (a.max(axis=0)[3],)), 'sort %.3f' % capi.stats()['ms_sort'], flush=True)
fb = tr().framebuffer(True)
print('fb checksum', float(fb.sum()))
