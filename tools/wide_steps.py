"""Steps per ray of W-wide collapses of the benchmark's tree (W = 2, 4, 6, 8), primary rays and their shadow rays: what an 8-wide
layout would buy in dependent steps, priced against what a node of that width costs to fetch and test (EXPERIMENTS.md).
   python tools/wide_steps.py [leaf_max=2]   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi, scenes
from gravit_amd.adapter import HipMeshAdapter
from oracle import orc  # ray generation only (a tool, not the product path)
capi.init(0)
for a in sys.argv[1:]:
    k, v = a.split("="); capi.set_option(k, int(v))
sc = scenes.soup_scene(10_000_000)
ad = HipMeshAdapter(sc.meshes[0])
c = sc.camera
rays = orc.camera_rays(c.eye, c.focus, c.up, c.fov, 1920, 1080)
nxt, t = orc.toplevel_intersect(sc.inst_lo, sc.inst_hi, [0], rays)
r = rays[nxt >= 0][::17]
o, d = np.ascontiguousarray(r["origin"]), np.ascontiguousarray(r["direction"])
vs = ad.visit_stats(o, d)
print("sample %d primary rays: binary inner visits %.1f, leaf visits %.2f, triangle tests %.2f per ray (closest hit)" % (len(o), vs["inner_per_ray"], vs["leaf_per_ray"], vs["tri_tests_per_ray"]))
for w in (2, 4, 6, 8):
    s = ad.wide_visit_stats(o, d, w)
    print("  %d-wide collapse: %8d nodes, %.1f node visits per ray (p99 %.0f, max %d)" % (w, s["wide_nodes"], s["nodes_per_ray"], s["p99"], s["max"]))
