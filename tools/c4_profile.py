import sys, time
sys.path.insert(0, '/root/repo')
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_SMOOTH
from gravit_amd.scheduler import ImageTracer
capi.init(0)
sc = scenes.bunny_grid_scene()
tr = ImageTracer(sc, NORMALS_SMOOTH)
for _ in range(3): tr()
capi.synchronize(); capi.stats_reset(); capi.profile(True)
n = 20
t = time.perf_counter()
for _ in range(n): tr()
capi.synchronize(); dt = (time.perf_counter() - t) / n
st = capi.stats(); capi.profile(False)
print("frame %.3f ms; per frame: closest %.3f long %.3f any %.3f shade %.3f shuffle %.3f; calls %d; launches closest %d any %d" % (
    dt * 1e3, st["ms_closest"] / n, st["ms_long"] / n, st["ms_any"] / n, st["ms_shade"] / n, st["ms_shuffle"] / n, tr.adapter_calls, st["launches_closest"] / n, st["launches_any"] / n))
