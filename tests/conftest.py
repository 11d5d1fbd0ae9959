import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built_product_library():
    """A fresh checkout has no libgvt_hip.so (build artefacts are not tracked): cross-compile it once (hipcc needs no GPU) so that the
    ABI tests have a library to load.  No-op when the library is up to date; building is not a fallback -- nothing computes without a GPU."""
    from gravit_amd import _build

    try:
        _build.build()
    except Exception as e:  # no hipcc on this machine: the tests that need the library report it
        print("libgvt_hip.so could not be built here:", e, file=sys.stderr)
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def ref_vectors():
    import json

    return json.load(open(os.path.join(GOLDEN, "ref_vectors.json")))


@pytest.fixture(scope="session")
def oracle_vectors():
    import json

    import numpy as np

    d = np.load(os.path.join(GOLDEN, "oracle_vectors.npz"))
    out = {k: d[k] for k in ("org", "dirs", "hits", "occluded")}
    out["fb_hashes"] = json.loads(str(d["fb_hashes"]))
    return out


@pytest.fixture(scope="session")
def hip():
    """The product library on a real device.  GPU tests FAIL (not skip) when it cannot be used."""
    from gravit_amd import capi

    capi.init(0)
    return capi


@pytest.fixture(autouse=True)
def _default_knobs(request):
    """Tuning knobs are process-wide state of the library: every GPU test starts from (and leaves) the defaults."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from gravit_amd import capi
    from oracle import orc

    from tests import helpers

    capi.set_option("defaults", 0)
    orc.set_skip_known_misses(False)  # the checker is STRICT -- the reference's hop-by-hop shuffleRays, which is also the library's default -- unless a test
    helpers.DEFAULT_RULE = "strict"   # asks for the restated shortcut (rule="shortcut", with the device's knob skip_known = 1)
    yield
    capi.set_option("defaults", 0)
    orc.set_skip_known_misses(False)
    helpers.DEFAULT_RULE = "asis"


def read_ppm(path):
    import numpy as np

    b = open(path, "rb").read()
    parts = b.split(b"\n", 3)
    w, h = map(int, parts[1].split())
    return np.frombuffer(parts[3], np.uint8).reshape(h, w, 3)
