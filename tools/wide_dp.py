"""Is a COST-OPTIMAL wide collapse of the LBVH worth building?  (go / no-go gate: <= 17 eight-wide visits per primary ray at
>= 6.5 children per node.)  Builds the benchmark's tree on the GPU, downloads the binary nodes, chooses the wide nodes' roots by dynamic
programming over the tree (tools/wide_dp.c: minimum SAH-expected wide-node visits under the W-children constraint, the recurrence of Ylitie
et al. 2017), and counts, with the library's own traversal, how many of those wide nodes the frame's primary rays and their shadow rays visit --
next to the greedy collapse the 4-wide layout is built with (gvt_hip_wide_visit_stats).
   python tools/wide_dp.py [tris=10000000] [leaf_max=2]      (GPU box)"""
import ctypes as C, os, subprocess, sys, time
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import numpy as np


def load_dp():
    so = os.path.join(HERE, "libwide_dp.so")
    src = os.path.join(HERE, "wide_dp.c")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src, "-lm"])
    lib = C.CDLL(so)
    lib.wide_dp.restype = C.c_int
    lib.wide_greedy.restype = C.c_int
    lib.ploc_rebuild.restype = C.c_int
    return lib


def dp_marks(lib, nodes, W):
    marks = np.zeros(len(nodes), np.uint8)
    stats = np.zeros(4 + 17, np.float64)
    rc = lib.wide_dp(nodes.ctypes.data_as(C.c_void_p), C.c_int64(len(nodes)), C.c_int(W), marks.ctypes.data_as(C.c_void_p), stats.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    return marks, stats


def greedy_marks(lib, nodes, W):
    marks = np.zeros(len(nodes), np.uint8)
    rc = lib.wide_greedy(nodes.ctypes.data_as(C.c_void_p), C.c_int64(len(nodes)), C.c_int(W), marks.ctypes.data_as(C.c_void_p))
    assert rc == 0, rc
    return marks


def main():
    from gravit_amd import capi, scenes
    from gravit_amd.adapter import HipMeshAdapter
    from oracle import orc  # ray generation only (a tool, not the product path)

    tris = 10_000_000
    capi.init(0)
    for a in sys.argv[1:]:
        k, v = a.split("=")
        if k == "tris": tris = int(v)
        else: capi.set_option(k, int(v))
    lib = load_dp()
    sc = scenes.soup_scene(tris)
    ad = HipMeshAdapter(sc.meshes[0])
    c = sc.camera
    rays = orc.camera_rays(c.eye, c.focus, c.up, c.fov, 1920, 1080)
    nxt, t = orc.toplevel_intersect(sc.inst_lo, sc.inst_hi, [0], rays)
    r = rays[nxt >= 0][::17]
    o, d = np.ascontiguousarray(r["origin"]), np.ascontiguousarray(r["direction"])
    vs = ad.visit_stats(o, d)
    print("soup of %d triangles, %d primary rays (every 17th of the frame's): binary inner visits %.1f, leaf visits %.2f, triangle tests %.2f per ray" % (
        tris, len(o), vs["inner_per_ray"], vs["leaf_per_ray"], vs["tri_tests_per_ray"]))
    t0 = time.time()
    nodes = ad.download_nodes()
    print("binary tree: %d inner nodes (%.1f s to download)" % (len(nodes), time.time() - t0))
    print("%-28s %10s %9s %10s %12s %8s %6s" % ("collapse", "wide nodes", "children", "SAH visits", "visits / ray", "p99", "max"))
    for W in (4, 6, 8):
        g = ad.wide_visit_stats(o, d, W)
        gm = greedy_marks(lib, nodes, W)
        cnt = ad.marked_visit_stats(o, d, gm)
        print("%-28s %10d %9s %10s %12.2f %8.0f %6d   (library's own marks: %.2f)" % ("greedy, %d-wide" % W, int(gm.sum()), "-", "-", cnt.mean(), np.percentile(cnt, 99), cnt.max(), g["nodes_per_ray"]))
        t0 = time.time()
        m, st = dp_marks(lib, nodes, W)
        cnt = ad.marked_visit_stats(o, d, m)
        fill = " ".join("%d:%.0f%%" % (k, 100 * st[4 + k] / st[0]) for k in range(2, W + 1))
        print("%-28s %10d %9.2f %10.2f %12.2f %8.0f %6d   (DP %.1f s; binary tree SAH %.1f; children histogram %s)" % (
            "cost-optimal (DP), %d-wide" % W, int(st[0]), st[1] / st[0], st[2], cnt.mean(), np.percentile(cnt, 99), cnt.max(), time.time() - t0, st[3], fill))
    # the restructuring step: the same leaves under a tree built by locally-ordered clustering (PLOC), then the same cost-optimal collapse
    for radius in (8, 32):
        t0 = time.time()
        out = np.zeros_like(nodes)
        rc = lib.ploc_rebuild(nodes.ctypes.data_as(C.c_void_p), C.c_int64(len(nodes)), C.c_int(radius), out.ctypes.data_as(C.c_void_p))
        assert rc == 0, rc
        ad.upload_nodes(out)
        vs2 = ad.visit_stats(o, d)
        print("PLOC tree over the same leaves, search radius %d (%.0f s on one host core): binary inner visits %.1f, leaf visits %.2f, triangle tests %.2f per ray" % (
            radius, time.time() - t0, vs2["inner_per_ray"], vs2["leaf_per_ray"], vs2["tri_tests_per_ray"]))
        for W in (4, 8):
            m, st = dp_marks(lib, out, W)
            cnt = ad.marked_visit_stats(o, d, m)
            fill = " ".join("%d:%.0f%%" % (k, 100 * st[4 + k] / st[0]) for k in range(2, W + 1))
            print("%-28s %10d %9.2f %10.2f %12.2f %8.0f %6d   (binary tree SAH %.1f; children histogram %s)" % (
                "  + cost-optimal, %d-wide" % W, int(st[0]), st[1] / st[0], st[2], cnt.mean(), np.percentile(cnt, 99), cnt.max(), st[3], fill))
            gm = greedy_marks(lib, out, W)
            cnt = ad.marked_visit_stats(o, d, gm)
            print("%-28s %10d %9s %10s %12.2f %8.0f %6d" % ("  + greedy, %d-wide" % W, int(gm.sum()), "-", "-", cnt.mean(), np.percentile(cnt, 99), cnt.max()))
    ad.upload_nodes(nodes)


if __name__ == "__main__":
    main()
