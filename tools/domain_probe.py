"""Rehearsal of the multi-rank path on ONE GPU: W in-process ranks (threads, own contexts) render the benchmark soup cut into W
domains through the native Domain scheduler.  The ranks share the GPU, so this shows protocol overhead and round counts, not scaling.
   python tools/domain_probe.py [world] [tris]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_FLAT
from gravit_amd.scheduler import Comm, Context, NativeTracer

capi.init(0)
world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
tris = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
sc = scenes.soup_domains_scene(tris, world)
owner = [i % world for i in range(sc.n_inst)]
for bsp in (False, True):
    hub = capi.load().gvt_hip_hub_create(world)
    bar = threading.Barrier(world)
    res = {}
    def work(rank):
        ctx = Context(0)
        comm = Comm.local(hub, rank)
        tr = NativeTracer(sc, NORMALS_FLAT, owner, comm)
        for _ in range(3):
            tr(bsp=bsp)
        bar.wait()
        t0 = time.perf_counter()
        for _ in range(10):
            tr(bsp=bsp)
        res[rank] = ((time.perf_counter() - t0) / 10 * 1e3, dict(tr.stats))
        tr.close(); comm.close(); tr = None; ctx.close()
    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    [t.start() for t in th]; [t.join() for t in th]
    capi.load().gvt_hip_hub_destroy(hub)
    print("world %d %s: %.3f ms/frame (max over ranks); rank 0: %s" % (world, "BSP" if bsp else "async", max(v[0] for v in res.values()), res[0][1]), flush=True)
