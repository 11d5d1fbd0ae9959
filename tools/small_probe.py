import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_SMOOTH, NORMALS_FLAT
from gravit_amd.scheduler import NativeTracer
capi.init(0)
cases = [("config 4", scenes.bunny_grid_scene(), NORMALS_SMOOTH), ("soup 10M x 8 domains", scenes.soup_domains_scene(10_000_000, 8), NORMALS_FLAT),
         ("simple 25 inst 512^2", scenes.simple_scene(512, 512), NORMALS_SMOOTH)]
for name, sc, mode in cases:
    tr = NativeTracer(sc, mode)
    for sr in (0, 512, 1024, 2048, 4096, 8192, 16384, 65536):
        capi.set_option("small_rays", sr)
        for _ in range(3): tr()
        t = time.perf_counter()
        for _ in range(10): tr()
        capi.synchronize()
        print("%-24s small_rays=%6d  %.3f ms/frame  chains %d" % (name, sr, (time.perf_counter() - t) / 10 * 1e3, tr.stats["chains"]), flush=True)
    tr.close()
