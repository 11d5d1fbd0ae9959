"""Where does the time of a 10 M-triangle acceleration-structure build go?  Builds the benchmark mesh a few times with GVT_HIP_BUILD_TRACE
(a stream synchronisation + a line per stage on stderr: the stages' sum is longer than an untraced build) and prints the untraced build_ms.
   usage (GPU box): python3 tools/build_probe.py [tris=10000000] [reps=3]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gravit_amd import capi, scenes
from gravit_amd.adapter import HipMeshAdapter

tris, reps, notrace = 10_000_000, 3, False
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k == "tris": tris = int(v)
    elif k == "reps": reps = int(v)
    elif k == "notrace": notrace = bool(int(v))
    else: capi.set_option(k, int(v))
capi.init(0)
mesh = scenes.soup_scene(tris).meshes[0]
for r in range(reps):
    ad = HipMeshAdapter(mesh)
    print("untraced build %d: %.3f ms  %s" % (r, ad.info()["build_ms"], {k: ad.info()[k] for k in ("n_nodes", "n_leaves") if k in ad.info()}), flush=True)
    ad.close()
if notrace:
    sys.exit(0)
os.environ["GVT_HIP_BUILD_TRACE"] = "1"
ad = HipMeshAdapter(mesh)
print("traced build: %.3f ms" % ad.info()["build_ms"], flush=True)
