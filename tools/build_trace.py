import os, sys
sys.path.insert(0, "/root/repo")
os.environ["GVT_HIP_BUILD_TRACE"] = "1"
from gravit_amd import capi, scenes
from gravit_amd.adapter import HipMeshAdapter
capi.init(0)
sc = scenes.soup_scene(10_000_000)
for k in range(3):
    ad = HipMeshAdapter(sc.meshes[0]); print("build_ms", ad.info()["build_ms"], flush=True); ad.close()
