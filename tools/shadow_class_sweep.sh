export GVT_HIP_LIB=$PWD/gravit_amd/libgvt_hip_exp.so
for cfg in "16 3" "24 3" "32 3" "24 2" "32 2" "40 2" "16 4" "0 4" "32 4"; do set -- $cfg
python - <<PY
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_FLAT
from gravit_amd.scheduler import NativeTracer
capi.init(0)
capi.set_option("shadow_cls_lo", $1); capi.set_option("shadow_cls_shift", $2)
sc = scenes.soup_scene(10_000_000)
tr = NativeTracer(sc, NORMALS_FLAT)
for _ in range(12): tr()
out=[]
for o in (0,1,0,1,0,1):
    capi.set_option("shadow_order", o)
    for _ in range(4): tr()
    capi.synchronize(); capi.stats_reset(); capi.profile(2)
    t=time.perf_counter()
    for _ in range(40): tr()
    capi.synchronize(); dt=(time.perf_counter()-t)/40*1e3
    st=capi.stats(); capi.profile(False)
    out.append((o, dt, st["ms_any"]/40))
a0=np.mean([x[2] for x in out if x[0]==0]); a1=np.mean([x[2] for x in out if x[0]==1])
f0=np.mean([x[1] for x in out if x[0]==0]); f1=np.mean([x[1] for x in out if x[0]==1])
print("lo=$1 shift=$2: any %.4f -> %.4f ms, frame %.4f -> %.4f; classes %s" % (a0,a1,f0,f1,capi.counters_peek()[24:32]), flush=True)
PY
done
