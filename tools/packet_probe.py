import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_FLAT, NORMALS_SMOOTH
from gravit_amd.scheduler import NativeTracer
capi.init(0)
for name, sc, mode in (("soup 10M", scenes.soup_scene(10_000_000), NORMALS_FLAT), ("soup 1M", scenes.soup_scene(1_000_000), NORMALS_FLAT),
                       ("bun_zipper", scenes.bunny70k_scene(), NORMALS_SMOOTH)):
    tr = NativeTracer(sc, mode)
    for pk in (0, 1):
        capi.set_option("packet", pk)
        for _ in range(3): tr()
        capi.stats_reset(); capi.profile(True)
        t = time.perf_counter()
        for _ in range(10): tr()
        capi.synchronize(); dt = (time.perf_counter() - t) / 10 * 1e3
        st = capi.stats(); capi.profile(False)
        print("%-12s packet=%d  frame %.3f ms  closest %.3f any %.3f  bailed packets %d of %d  %s" % (name, pk, dt, st["ms_closest"] / 10, st["ms_any"] / 10,
              tr.stats["packets_bailed"], (tr.stats["rays_closest"] + 63) // 64 * 2, ""), flush=True)
    tr.close()
