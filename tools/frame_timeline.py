"""Kernel timeline of ONE benchmark frame (native tracer, 10 M soup, 1080p) from a rocprofv3 kernel trace.
   GPU box:  cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/tl -o tl -- python3 $REPO/tools/frame_timeline.py run
   then:     python tools/frame_timeline.py parse gpurun_out/tl/.../tl_kernel_trace.csv"""
import csv, glob, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    from gravit_amd import capi, scenes
    from gravit_amd.layouts import NORMALS_FLAT
    from gravit_amd.scheduler import NativeTracer
    capi.init(0)
    n_dom = 1
    for a in sys.argv[2:]:
        k, v = a.split("=")
        if k == "domains": n_dom = int(v)
        else: capi.set_option(k, int(v))
    tr = NativeTracer(scenes.soup_scene(10_000_000) if n_dom == 1 else scenes.soup_domains_scene(10_000_000, n_dom), NORMALS_FLAT)
    for _ in range(6 if n_dom == 1 else 60):  # (several domains: the tracer first times its alternatives for the small rounds)
        tr()
    capi.synchronize()
else:
    path = sys.argv[2]
    if os.path.isdir(path):
        path = glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # frames start with the framebuffer clear (the largest fill); take the last complete one
    def name(r):
        m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"])
        return m.group(1) if m else r["Kernel_Name"][:32]
    # a frame starts at the framebuffer clear (fill) or the lean path's k_cam1_count and ends with its last k_round_report
    starts = [i for i, r in enumerate(rows) if name(r) in ("k_cam1_count", "k_zero_totals")]
    a = starts[-2] if len(starts) > 1 else starts[-1]
    nxt = [i for i in starts if i > a]
    b = (nxt[0] if nxt else len(rows)) - 1
    while b > a and name(rows[b]) != "k_round_report":
        b -= 1
    t0 = int(rows[a]["Start_Timestamp"])
    prev_end = t0
    busy = 0
    for r in rows[a:b + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        print("%9.1f us  +%6.1f gap  %8.1f us  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, name(r)))
        busy += e - s
        prev_end = max(prev_end, e)
    print("frame: %d kernels, %.1f us from first start to last end, %.1f us inside kernels" % (b + 1 - a, (prev_end - t0) / 1e3, busy / 1e3))
