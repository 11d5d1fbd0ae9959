"""What do k compute units reserved for the communicator's stream (knob comm_cus) cost the traversal?  (the side one GPU can measure.)
For k in 0 / 8 / 16 / 32: a fresh context, comm_cus = k, a one-rank in-process communicator created on it (gvt_hip_comm_create_local: that is where the CU masks are
applied -- the communicator's own stream gets the first k CUs, the context's compute stream the others, the persistent grids are sized for the rest), then the
benchmark frame (10 M-triangle soup, 1080p, one domain) and the same soup in 8 domains on that context, alternated over `blocks` blocks of `frames` frames.
   python tools/comm_cus_probe.py [tris=10000000] [blocks=3] [frames=30]      (GPU box)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_FLAT
from gravit_amd.scheduler import Comm, Context, NativeTracer

opt = dict(a.split("=") for a in sys.argv[1:])
tris, blocks, frames = int(opt.get("tris", 10_000_000)), int(opt.get("blocks", 3)), int(opt.get("frames", 30))
capi.init(0)
sc1 = scenes.soup_scene(tris)
sc8 = scenes.soup_domains_scene(tris, 8)
res = {}
for b in range(blocks):
    for k in (0, 8, 16, 32):
        ctx = Context(0)
        capi.set_option("comm_cus", k)
        hub = capi.load().gvt_hip_hub_create(1)
        comm = Comm.local(hub, 0)
        assert comm.reserved_cus == k
        for name, sc in (("one domain", sc1), ("8 domains", sc8)):
            tr = NativeTracer(sc, NORMALS_FLAT)
            for _ in range(14):
                tr()
            capi.synchronize()
            t = time.perf_counter()
            for _ in range(frames):
                tr()
            capi.synchronize()
            res.setdefault((name, k), []).append((time.perf_counter() - t) / frames * 1e3)
            tr.close()
        comm.close(); capi.load().gvt_hip_hub_destroy(hub)
        import gc; gc.collect()
        ctx.close()
for name in ("one domain", "8 domains"):
    base = float(np.mean(res[(name, 0)]))
    for k in (0, 8, 16, 32):
        v = res[(name, k)]
        print("%-11s comm_cus = %2d: %.4f ms per frame (min %.4f, max %.4f over %d blocks of %d frames)  %+.1f %%" % (name, k, float(np.mean(v)), min(v), max(v), blocks, frames, (float(np.mean(v)) / base - 1) * 100))
