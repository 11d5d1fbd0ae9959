"""Phase breakdown of k_trace from s_memtime stamps, with a private -DGVT_STAMP=1 build of the library.
   usage: python tools/stamps.py --build            (here, cross-compiles tools/libgvt_hip_stamp.so)
          python tools/stamps.py [ntris] [opt=value ...]   (on the GPU box)"""
import sys, ctypes, subprocess
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi, scenes, _build
STAMP_LIB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgvt_hip_stamp.so")
if "--build" in sys.argv:
    srcs = [os.path.join(_build.CSRC, s) for s in _build.SOURCES]
    level = [a for a in sys.argv if a.startswith("--level=")]
    subprocess.check_call([_build.hipcc()] + _build.FLAGS + ["-DGVT_STAMP=" + (level[0][8:] if level else "1"), "-DGVT_EXPERIMENTS", "-shared", "-o", STAMP_LIB] + srcs)
    print(STAMP_LIB); sys.exit(0)
capi.LIB_PATH = STAMP_LIB
from gravit_amd.scheduler import ImageTracer
capi.init(0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
for a in sys.argv[2:]:
    k, v = a.split("="); capi.set_option(k, int(v))
sc = scenes.soup_scene(N)
tr = ImageTracer(sc, 0)
tr(); capi.synchronize()
buf = (ctypes.c_ulonglong * 24)()
lib = capi.load()
lib.gvt_hip_debug_stamps.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
lib.gvt_hip_debug_stamps(buf, 1)
tr(); capi.synchronize()
lib.gvt_hip_debug_stamps(buf, 0)
v = list(buf)
names = ["refill", "inner", "leaf", "retire", "inner_iters", "outer_iters", "waves", "wave_cycles", "drain_cycles", "drain_inner", "drain_outer", "act_at_exh", "max_drain_cyc", "max_drain_inner"]
d = dict(zip(names, v))
print(d)
w = max(1, d["waves"])
print("per wave: cycles %.0f  refill %.0f inner %.0f leaf %.0f retire %.0f | inner iters %.1f (%.0f cyc each) outer iters %.1f (leaf phase %.0f cyc each)" % (
    d["wave_cycles"] / w, d["refill"] / w, d["inner"] / w, d["leaf"] / w, d["retire"] / w, d["inner_iters"] / w, d["inner"] / max(1, d["inner_iters"]),
    d["outer_iters"] / w, d["leaf"] / max(1, d["outer_iters"])))
