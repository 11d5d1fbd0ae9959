import sys, os
sys.path.insert(0, os.getcwd())
from gravit_amd import capi
import tests.helpers as H
H.DEFAULT_RULE = "strict"
import tests.test_gpu_parity as P
capi.init(0)
class Hip:
    set_option = staticmethod(capi.set_option); stats = staticmethod(capi.stats)
bad = 0
for s in [2044, 3358, 3662, 4948, 6715, 8347] + list(range(10600, 12200)):
    try:
        P.test_random_meshes_and_rays_against_the_oracle(Hip, s)
    except Exception as e:
        bad += 1
        import traceback
        print("FAIL", s, traceback.format_exc()[-600:], flush=True)
        if bad > 6: break
print("done, failures:", bad)
