"""Occupancy sweep of the traversal launches (resident blocks per CU) -- run it against builds with other LDS stack depths:
   GVT_EXTRA_HIPCC_FLAGS=-DTRAV_STACK=12 python -m gravit_amd._build && cp gravit_amd/libgvt_hip.so /tmp/s12.so && GVT_HIP_LIB=/tmp/s12.so python tools/stack_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_FLAT
from gravit_amd.scheduler import NativeTracer
capi.init(0)
sc = scenes.soup_scene(10_000_000)
tr = NativeTracer(sc, NORMALS_FLAT)
for bc in (4, 5, 6, 7, 8):
    for ba in (4, 6, 8):
        capi.set_option("defaults", 0); capi.set_option("blocks_per_cu_closest", bc); capi.set_option("blocks_per_cu", ba)
        for _ in range(3): tr()
        capi.stats_reset(); capi.profile(2)
        for _ in range(10): tr()
        st = capi.stats(); capi.profile(0)
        print("closest blocks/CU %d any blocks/CU %d: closest %.3f long %.3f any %.3f" % (bc, ba, st["ms_closest"] / 10, st["ms_long"] / 10, st["ms_any"] / 10), flush=True)
