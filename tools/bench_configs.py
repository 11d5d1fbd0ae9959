"""Frame time and ray rate of the BASELINE.json configurations on ONE GPU through the native tracer (rounds; gvt_hip_tracer) and, for
comparison, through the reference-order loop (one adapter call at a time, gvt_hip_image_frame).
   usage (GPU box): python tools/bench_configs.py [only=1,2,5] [roofline=1] [opt=value ...]
roofline=1 adds, per configuration, the per-class kernel times of the round chain (HIP events on the launch stream, gvt_hip_profile(2)), the dominant
traversal class, its algorithmic bytes per ray by SURVEY 8(d)'s formula at the configuration's triangle count (the largest mesh of the scene) and the
fraction of the 8 TB/s HBM roofline that launch class reaches; plus the builder's per-mesh packet statistic."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_SMOOTH, NORMALS_FLAT
from gravit_amd.scheduler import ImageTracer, NativeTracer

capi.init(0)
import math
ONLY, ROOF, FRAMES, NOREF = None, False, 10, False
for a in sys.argv[1:]:  # opt=value: library options; only=1,2,4: a subset of the configurations; frames=N timed frames; noref=1: skip the reference-order loop
    k, v = a.split("=")
    if k == "only": ONLY = set(v.split(","))
    elif k == "roofline": ROOF = bool(int(v))
    elif k == "frames": FRAMES = int(v)
    elif k == "noref": NOREF = bool(int(v))
    else: capi.set_option(k, int(v))


def b_ray(n_tris):
    return 32 + 16 + 32 * max(1, math.ceil(math.log2(max(n_tris, 8) / 4.0))) + 4 * 48
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def config5(size, n_dom):
    one = scenes.cathedral_scene(size, size, samples=2, depth=2, eye=(0.0, 1.5, 13.0), light=(0.0, 2.5, 12.0))
    return one if n_dom <= 1 else scenes.split_into_domains(one, n_dom)


cfgs = [
    ("config 1: data/bunny.conf, 3 bunny instances, 1900x1080", lambda: scenes.load_conf(os.path.join(GOLDEN, "bunny.conf")), NORMALS_SMOOTH),
    ("config 2: bun_zipper (69 K tris), 1920x1080", lambda: scenes.bunny70k_scene(), NORMALS_SMOOTH),
    ("config 3: 10 M-triangle soup, 1920x1080", lambda: scenes.soup_scene(10_000_000), NORMALS_FLAT),
    ("config 4: 8-instance bunny grid, 1900x1080", lambda: scenes.bunny_grid_scene(), NORMALS_SMOOTH),
    ("config 5 stand-in: hall 80 K tris, 1024x1024, 2x2 samples, depth 2", lambda: config5(1024, 1), NORMALS_FLAT),
    ("config 5 stand-in cut into 8 domains (one rank)", lambda: config5(1024, 8), NORMALS_FLAT),
]
for name, mk, mode in cfgs:
    if ONLY is not None and name.split(":")[0].replace("config ", "").split(" ")[0] not in ONLY:
        continue
    sc = mk()
    tris = sum(len(m.tris) for m in sc.meshes)
    for label, mk_tr in (("rounds", lambda: NativeTracer(sc, mode)), ("reference order", lambda: ImageTracer(sc, mode))):
        if NOREF and label != "rounds":
            continue
        tr = mk_tr()
        for _ in range(30):  # (warm-up; the tracer's per-scene choices -- parking threshold, the frame's route: k_finish or per-hop rounds, hops never / early / always -- settle within 20 frames)
            tr()
        capi.synchronize(); capi.stats_reset()
        n = FRAMES
        t = time.perf_counter()
        for _ in range(n):
            tr()
        capi.synchronize(); dt = (time.perf_counter() - t) / n
        st = capi.stats()
        rays = (st["rays_closest"] + st["rays_any"]) / n
        extra = ("%d launch chains, %d host syncs" % (tr.stats["chains"], tr.stats["host_syncs"])) if label == "rounds" else ("%d adapter calls" % tr.adapter_calls)
        print("%-70s %-16s %8.3f ms/frame %8.1f Mrays/s  (%d rays/frame, %s, %d tris, %d instances)" % (
            name, label, dt * 1e3, rays / dt / 1e6, rays, extra, tris, sc.n_inst), flush=True)
        if ROOF and label == "rounds":
            capi.stats_reset(); capi.profile(2)
            for _ in range(n):
                tr()
            st = capi.stats(); capi.profile(False)
            dom = "closest" if st["ms_closest"] >= st["ms_any"] else "any"
            T = max(len(m.tris) for m in sc.meshes)
            B = b_ray(T)
            rays_dom = st["rays_%s" % dom]
            ms_dom = st["ms_%s" % dom]
            ach = rays_dom * B / (ms_dom * 1e-3) / 1e9 if ms_dom > 0 else 0.0
            infos = [a.info() for a in tr.backend.adapter_cache.values()]
            print("    kernel classes per frame: closest %.3f ms (%d launches), long %.3f ms, any %.3f ms (%d launches); dominant: %s-hit, %d rays per frame through it, "
                  "%d algorithmic B/ray at T = %d -> %.0f GB/s = %.3f of the 8 TB/s roofline; packet statistic per mesh: %s" % (
                      st["ms_closest"] / n, st["launches_closest"] // n, st["ms_long"] / n, st["ms_any"] / n, st["launches_any"] // n, dom, rays_dom // n, B, T, ach, ach / 8000.0,
                      ", ".join(sorted(set("sah %.0f -> %s" % (i["sah_inner"], "packets" if i["packet"] else "lanes") for i in infos)))), flush=True)
        if hasattr(tr, "close"):
            tr.close()
