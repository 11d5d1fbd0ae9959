export GVT_HIP_LIB=$PWD/gravit_amd/libgvt_hip_exp.so
python - <<PY
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_FLAT
from gravit_amd.scheduler import NativeTracer
capi.init(0)
sc = scenes.soup_scene(10_000_000)
tr = NativeTracer(sc, NORMALS_FLAT)
for _ in range(12): tr()
def run(opts, frames=40):
    capi.set_option("defaults", 0)
    for k, v in opts.items(): capi.set_option(k, v)
    for _ in range(4): tr()
    capi.synchronize(); capi.stats_reset(); capi.profile(2)
    t = time.perf_counter()
    for _ in range(frames): tr()
    capi.synchronize(); dt = (time.perf_counter() - t) / frames * 1e3
    st = capi.stats(); capi.profile(False)
    return dt, st["ms_closest"] / frames, st["ms_any"] / frames
cases = [{}, {"share": 0}, {"refill_min": 8}, {"refill_min": 32}, {"inner_min": 16}, {"inner_min": 48}, {"share_min_rays": 1 << 30}, {"long_steps": 80}, {"long_steps": 112}, {}]
for o in cases:
    r = [run(o) for _ in range(2)]
    print("%-28s frame %.4f  closest %.4f  any %.4f" % (o, min(x[0] for x in r), min(x[1] for x in r), min(x[2] for x in r)), flush=True)
PY
