import sys, time, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi, scenes
from gravit_amd.scheduler import ImageTracer
capi.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sc = scenes.soup_scene(N)
tr = ImageTracer(sc, 0)
def run(opts, frames=6):
    for k, v in opts.items(): capi.set_option(k, v)
    tr(); capi.synchronize()
    capi.stats_reset(); capi.profile(True)
    t = time.perf_counter()
    for _ in range(frames): tr()
    capi.synchronize(); dt = (time.perf_counter() - t) / frames
    st = capi.stats(); capi.profile(False)
    fb = tr().framebuffer(True).copy()
    return (dt * 1e3, st['ms_closest'] / frames, st['ms_any'] / frames, (st['rays_closest'] + st['rays_any']) / frames / dt / 1e6), fb
ref = None
for opts in (dict(term_sink=1), dict(term_sink=0), dict(term_sink=1), dict(term_sink=0)):
    r, fb = run(opts, frames=12)
    if ref is None: ref = fb
    st = capi.stats()
    print(json.dumps(opts), 'frame %.3f ms closest %.3f long %.3f any %.3f Mrays/s %.0f' % (r[0], r[1], st['ms_long'] / 12, r[2], r[3]), 'shuffle %.3f' % (st['ms_shuffle'] / 12), 'fb equal:', bool(np.array_equal(fb, ref)), flush=True)
