"""Soak of the native Domain scheduler on in-process ranks: many frames, every frame's composited image compared with the first.
   python tools/soak_domain.py [world] [frames] [option=value ...]     (library options, set in every rank's context)"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_FLAT, NORMALS_SMOOTH
from gravit_amd.scheduler import Comm, Context, NativeTracer

capi.init(0)
pos = [a for a in sys.argv[1:] if "=" not in a]
lib_opts = [(a.split("=")[0], int(a.split("=")[1])) for a in sys.argv[1:] if "=" in a]
world = int(pos[0]) if len(pos) > 0 else 4
frames = int(pos[1]) if len(pos) > 1 else 100
one = scenes.cathedral_scene(256, 256, samples=2, depth=2, eye=(0.0, 1.5, 13.0), light=(0.0, 2.5, 12.0))
cases = [("config 4", scenes.bunny_grid_scene(width=380, height=216), NORMALS_SMOOTH, 0.0), ("config 5 x8", scenes.split_into_domains(one, 8), NORMALS_FLAT, 1e-5)]
for name, sc, mode, tol in cases:
    owner = [i % world for i in range(sc.n_inst)]
    for bsp in (False, True):
        hub = capi.load().gvt_hip_hub_create(world)
        worst, errs = [0.0], []
        def work(rank):
            try:
                ctx = Context(0)
                for k, v in lib_opts:
                    capi.set_option(k, v)
                comm = Comm.local(hub, rank)
                tr = NativeTracer(sc, mode, owner, comm)
                first = None
                for f in range(frames):
                    B = tr(bsp=bsp)
                    if rank == 0:
                        fb = B.framebuffer(True)
                        if first is None:
                            first = fb.copy()
                        else:
                            worst[0] = max(worst[0], float(np.abs(fb - first).max()))
                            assert np.array_equal(fb[..., 3], first[..., 3]), "deposit counts changed in frame %d" % f
                tr.close(); comm.close(); tr = B = None; ctx.close()
            except Exception:
                import traceback
                errs.append(traceback.format_exc()); capi.load().gvt_hip_hub_abort(hub)
        t0 = time.time()
        th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
        [t.start() for t in th]; [t.join() for t in th]
        capi.load().gvt_hip_hub_destroy(hub)
        assert not errs, errs[0]
        assert worst[0] <= tol, (name, worst[0])
        print("%s, %d ranks, %s: %d frames, max deviation from the first frame %.3g (%.1f s)" % (name, world, "BSP" if bsp else "async", frames, worst[0], time.time() - t0), flush=True)
print("soak ok")
