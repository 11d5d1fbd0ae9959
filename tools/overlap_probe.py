"""How much would overlapping launches buy?  Two contexts (threads, own streams) render the benchmark frame concurrently:
python tools/overlap_probe.py   (GPU box)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_FLAT
from gravit_amd.scheduler import Context, NativeTracer

capi.init(0)
sc = scenes.soup_scene(10_000_000)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 2
trs, ready, go = [None] * K, threading.Barrier(K + 1), threading.Barrier(K + 1)
times = [0.0] * K
def work(k):
    ctx = Context(0)
    tr = NativeTracer(sc, NORMALS_FLAT)
    for _ in range(3):
        tr()
    ready.wait(); go.wait()
    t0 = time.perf_counter()
    for _ in range(20):
        tr()
    times[k] = time.perf_counter() - t0
    tr.close(); tr = None; ctx.close()
th = [threading.Thread(target=work, args=(k,)) for k in range(K)]
[t.start() for t in th]
ready.wait()
t0 = time.perf_counter()
go.wait()
[t.join() for t in th]
wall = time.perf_counter() - t0
print("%d concurrent contexts: %.3f ms per frame each, %.3f ms per frame aggregate" % (K, max(times) / 20 * 1e3, wall / (20 * K) * 1e3))
