"""Would cutting ONE frame into concurrently traced slices pay?  K contexts (threads, own streams) each render the benchmark scene at
1920 x (1080 / K): together the rays of one 1080p frame.  Compared with one context rendering the whole frame.
   python tools/slice_probe.py [K=2]   (GPU box)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_FLAT
from gravit_amd.scheduler import Context, NativeTracer

capi.init(0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
full = scenes.soup_scene(10_000_000)
tr = NativeTracer(full, NORMALS_FLAT)
for _ in range(3): tr()
t0 = time.perf_counter()
for _ in range(20): tr()
one = (time.perf_counter() - t0) / 20 * 1e3
rays_full = tr.stats["rays_closest"] + tr.stats["rays_any"]
part = scenes.soup_scene(10_000_000, 1920, 1080 // K)
part.meshes = full.meshes
ready, go = threading.Barrier(K + 1), threading.Barrier(K + 1)
rays = [0] * K
def work(k):
    ctx = Context(0)
    t = NativeTracer(part, NORMALS_FLAT, backend=None)
    for _ in range(3): t()
    ready.wait(); go.wait()
    for _ in range(20):
        t()
    rays[k] = t.stats["rays_closest"] + t.stats["rays_any"]
    t.close(); t = None; ctx.close()
th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
[t.start() for t in th]
ready.wait(); t0 = time.perf_counter(); go.wait()
[t.join() for t in th]
wall = (time.perf_counter() - t0) / 20 * 1e3
print("one context, 1920x1080: %.3f ms per frame (%d rays); %d contexts, 1920x%d each, concurrently: %.3f ms per round (%d rays together)" % (one, rays_full, K, 1080 // K, wall, sum(rays)))
