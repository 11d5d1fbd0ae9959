"""Timing probe for the round chain on the benchmark scene: python tools/fused_probe.py [tris]  (GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_FLAT
from gravit_amd.scheduler import NativeTracer

capi.init(0)
tris = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
sc = scenes.soup_scene(tris)
tr = NativeTracer(sc, NORMALS_FLAT)
sc0 = scenes.soup_scene(1000)  # same camera, no lights variant built below
def run(tag, opts, tracer=tr, steps=10):
    capi.set_option("defaults", 0)
    for k, v in opts.items():
        capi.set_option(k, v)
    for _ in range(3):
        tracer()
    capi.stats_reset(); capi.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        tracer()
    capi.synchronize()
    dt = (time.perf_counter() - t0) / steps * 1e3
    st = capi.stats(); capi.profile(False)
    print("%-40s frame %.3f ms  closest %.3f any %.3f shade %.3f shuffle %.3f long %.3f" % (
        tag, dt, st["ms_closest"] / steps, st["ms_any"] / steps, st["ms_shade"] / steps, st["ms_shuffle"] / steps, st["ms_long"] / steps), flush=True)
run("default (packets)", dict())
run("packet=0", dict(packet=0))
run("packet=1 blocks irrelevant, camera_tile=0", dict(camera_tile=0))
