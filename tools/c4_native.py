import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_SMOOTH
from gravit_amd.scheduler import ImageTracer
capi.init(0)
sc = scenes.bunny_grid_scene()
for native in (True, False, True, False):
    tr = ImageTracer(sc, NORMALS_SMOOTH, native=native)
    for _ in range(3): tr()
    capi.synchronize()
    n = 30
    t = time.perf_counter()
    for _ in range(n): tr()
    capi.synchronize(); dt = (time.perf_counter() - t) / n
    print("native loop" if native else "python loop", "%.3f ms/frame, %d adapter calls" % (dt * 1e3, tr.adapter_calls))
