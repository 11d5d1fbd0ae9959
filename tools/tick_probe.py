"""Fixed cost of one exchange tick of the native Domain scheduler: two in-process ranks (threads, one context each, the library's
in-process transport) on ONE GPU render a toy scene whose rays cross from domain to domain several times -- compute is negligible, what is
left per tick is the scheduler's own cost: small chain, report, announce, payload.
   python tools/tick_probe.py [frames=300] [opt=value ...]      (GPU box)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_FLAT
from gravit_amd.scheduler import Comm, Context, NativeTracer

frames, opts, bsp = 300, [], False
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k == "frames": frames = int(v)
    elif k == "bsp": bsp = bool(int(v))
    else: opts.append((k, int(v)))
capi.init(0)
sc = scenes.soup_domains_scene(30000, 4, 96, 54)
sc.camera.eye, sc.camera.focus = (3.0, 0.6, 0.4), (0.5, 0.5, 0.5)  # along -x: rays cross the x-tiled domains one after the other
sc.lights["position"] = (2.0, 2.5, 1.5)
N = 2
owner = [i % N for i in range(sc.n_inst)]
hub = capi.load().gvt_hip_hub_create(N)
bar = threading.Barrier(N)
res, errs = {}, []

def rank_main(r):
    try:
        ctx = Context(0)
        for k, v in opts:
            capi.set_option(k, v)
        comm = Comm.local(hub, r)
        tr = NativeTracer(sc, NORMALS_FLAT, owner, comm)
        for _ in range(20):
            tr(bsp=bsp)
        capi.synchronize(); bar.wait()
        sums = {}
        t0 = time.perf_counter()
        for _ in range(frames):
            tr(bsp=bsp)
            for k, v in tr.stats.items():
                sums[k] = sums.get(k, 0) + v
        capi.synchronize(); bar.wait()
        res[r] = (sums, time.perf_counter() - t0)
        tr.close(); comm.close(); ctx.close()
    except Exception:
        import traceback
        errs.append(traceback.format_exc()); capi.load().gvt_hip_hub_abort(hub); bar.abort()

th = [threading.Thread(target=rank_main, args=(r,)) for r in range(N)]
[t.start() for t in th]; [t.join() for t in th]
if errs:
    print(errs[0]); sys.exit(1)
el = max(v[1] for v in res.values())
mx = lambda k: max(v[0].get(k, 0) for v in res.values())
ticks = mx("rounds") / frames
print("toy scene, 2 in-process ranks, %s, %d frames, options %s" % ("BSP" if bsp else "asynchronous", frames, opts))
print("  %.1f us per frame, %.2f ticks per frame -> %.1f us per tick; %.1f chains, %.1f host syncs, %d rays sent per frame, %d rays traced per frame" % (
    el / frames * 1e6, ticks, el / frames * 1e6 / ticks, mx("chains") / frames, mx("host_syncs") / frames, sum(v[0].get("rays_sent", 0) for v in res.values()) / frames,
    sum(v[0].get("rays_closest", 0) + v[0].get("rays_any", 0) for v in res.values()) / frames))
if any(k == "frame_timing" and v for k, v in opts):
    print("  per tick (max over ranks), us: " + ", ".join("%s %.1f" % (k[3:], mx(k) / frames / ticks * 1e3) for k in ("ms_chain", "ms_announce", "ms_payload", "ms_host_wait")) +
          "; composite per frame %.1f us" % (mx("ms_composite") / frames * 1e3))
