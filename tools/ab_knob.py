"""A/B of one library option on the benchmark frame inside ONE process: the native tracer on the 10 M-triangle soup at 1080p, the option's two (or more)
values alternated block by block so that drift and the box cancel.   usage (GPU box): python tools/ab_knob.py shadow_order 0 1 [blocks=6] [frames=40]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_FLAT
from gravit_amd.scheduler import NativeTracer

name = sys.argv[1]
vals = [int(v) for v in sys.argv[2:] if "=" not in v]
opt = dict(a.split("=") for a in sys.argv[2:] if "=" in a)
blocks, frames = int(opt.get("blocks", 6)), int(opt.get("frames", 40))
capi.init(0)
sc = scenes.soup_scene(10_000_000)
tr = NativeTracer(sc, NORMALS_FLAT)
for _ in range(12):
    tr()
res = {v: [] for v in vals}
cls = {v: {} for v in vals}
for b in range(blocks):
    for v in vals:
        capi.set_option(name, v)
        for _ in range(4):
            tr()
        capi.synchronize(); capi.stats_reset(); capi.profile(2)
        t = time.perf_counter()
        for _ in range(frames):
            tr()
        capi.synchronize()
        dt = (time.perf_counter() - t) / frames * 1e3
        st = capi.stats(); capi.profile(False)
        res[v].append(dt)
        for k in ("ms_closest", "ms_any", "ms_long", "ms_shade"):
            cls[v].setdefault(k, []).append(st[k] / frames)
        peek = capi.counters_peek()
        cls[v]["classes"] = peek[24:32]
for v in vals:
    print("%s=%d: %.4f ms per frame (min %.4f, max %.4f over %d blocks of %d frames); %s; shadow classes of the last frame %s" % (
        name, v, float(np.mean(res[v])), min(res[v]), max(res[v]), blocks, frames,
        ", ".join("%s %.4f" % (k[3:], float(np.mean(x))) for k, x in cls[v].items() if k != "classes"), cls[v]["classes"]))
