"""More seeds of the three seeded fuzz tests than the suite runs (tests/test_gpu_parity.py, tests/test_gpu_native.py): prints the first failure.
   usage (GPU box): python tools/fuzz_soak.py [queries=400] [calls=200] [scenes=60] [film=1] [option=value ...]"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from gravit_amd import capi
from oracle import orc
import tests.test_gpu_parity as P
import tests.test_gpu_native as N

class Hip:  # the fixture's surface the tests use
    set_option = staticmethod(capi.set_option)
    stats = staticmethod(capi.stats)

n = {"queries": 400, "calls": 200, "scenes": 60, "film": 1}  # film=k: the random scenes' films k times wider and higher (rounds beyond the small-round kernels)
lib_opts = {}  # any other name=value: a library option, set again before every seed (e.g. shadow_order_min_rays=0 long_min_rays=0: the class-ordered shadow list on small frames)
for a in sys.argv[1:]:
    k, v = a.split("=")
    if k in n: n[k] = int(v)
    else: lib_opts[k] = int(v)
if n["film"] > 1:
    _base = N._random_scene
    def _big(seed):
        sc = _base(seed)
        sc.camera.width *= n["film"]; sc.camera.height *= n["film"]
        return sc
    N._random_scene = _big
capi.init(0)
import tests.helpers as _H
_H.DEFAULT_RULE = "strict"  # as tests/conftest.py does for the GPU tests: the STRICT checker (the reference's hop-by-hop shuffle rule, the library's default)
bad = 0
for name, fn, cnt, base in (("queries", P.test_random_meshes_and_rays_against_the_oracle, n["queries"], 100), ("calls", P.test_random_adapter_calls_against_the_oracle, n["calls"], 100),
                            ("scenes", N.test_random_scenes_through_the_native_schedulers, n["scenes"], 100)):
    for s in range(base, base + cnt):
        try:
            for k_, v_ in lib_opts.items():
                capi.set_option(k_, v_)
            fn(Hip, s)
        except Exception:  # noqa: BLE001
            bad += 1
            print("FAIL %s seed %d\n%s" % (name, s, traceback.format_exc()[-1500:]), flush=True)
            if bad > 5: sys.exit(1)
    print("%s: %d seeds done, %d failures so far" % (name, cnt, bad), flush=True)
sys.exit(1 if bad else 0)
