"""Where a BVH build's time goes (GVT_HIP_BUILD_TRACE=1: a synchronisation and a line per stage): the 10 M-triangle soup built three times --
the first build of a process carries the loading of the kernels.   usage (GPU box): python3 tools/build_trace.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GVT_HIP_BUILD_TRACE"] = "1"
from gravit_amd import capi, scenes
from gravit_amd.adapter import HipMeshAdapter
capi.init(0)
sc = scenes.soup_scene(10_000_000)
for k in range(3):
    ad = HipMeshAdapter(sc.meshes[0]); print("build_ms", ad.info()["build_ms"], flush=True); ad.close()
