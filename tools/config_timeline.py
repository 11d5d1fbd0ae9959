"""Kernel timeline of ONE frame of a BASELINE configuration through the native tracer (rocprofv3 kernel trace).
   GPU box:  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o tl -- python3 $REPO/tools/config_timeline.py run <config> [opt=value ...]
             python3 tools/config_timeline.py parse OUT        (config: 1 = bunny.conf, 4 = bunny grid, 5 = hall, 58 = hall in 8 slabs, d8 = the benchmark soup in 8 domains)"""
import csv, glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    from gravit_amd import capi, scenes
    from gravit_amd.layouts import NORMALS_FLAT, NORMALS_SMOOTH
    from gravit_amd.scheduler import NativeTracer
    capi.init(0)
    cfg = sys.argv[2]
    for a in sys.argv[3:]:
        k, v = a.split("="); capi.set_option(k, int(v))
    golden = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    hall = lambda n: (lambda one: one if n <= 1 else scenes.split_into_domains(one, n))(scenes.cathedral_scene(1024, 1024, samples=2, depth=2, eye=(0.0, 1.5, 13.0), light=(0.0, 2.5, 12.0)))
    sc, mode = {"1": lambda: (scenes.load_conf(os.path.join(golden, "bunny.conf")), NORMALS_SMOOTH), "4": lambda: (scenes.bunny_grid_scene(), NORMALS_SMOOTH),
                "5": lambda: (hall(1), NORMALS_FLAT), "58": lambda: (hall(8), NORMALS_FLAT),
                "d8": lambda: (scenes.soup_domains_scene(10_000_000, 8, 1920, 1080), NORMALS_FLAT)}[cfg]()   # bench.py --domains 8
    tr = NativeTracer(sc, mode)
    for _ in range(24):  # (the frame's route has settled by then)
        tr()
    capi.synchronize()
else:
    path = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: (re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"]) or [None, r["Kernel_Name"][:32]])[1]
    starts = [i for i, r in enumerate(rows) if name(r) in ("k_cam1_count", "k_zero_totals")]
    a, b = starts[-2], starts[-1]
    t0, prev, busy = int(rows[a]["Start_Timestamp"]), int(rows[a]["Start_Timestamp"]), 0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%9.1f us  +%6.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, name(r)))
        busy += e - s; prev = max(prev, e)
    print("frame: %d kernels, %.1f us from first start to the next frame's start, %.1f us inside kernels" % (b - a, (int(rows[b]["Start_Timestamp"]) - t0) / 1e3, busy / 1e3))
