"""The knob sweeps over the variants that live in the experiments build only (libgvt_hip_exp.so): ONE child process runs
tests/experiment_cases.py with GVT_HIP_LIB pointing at that library (a process holds one build of the library), and the shipped
library's refusal of those knobs is checked here."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_experimental_variants_return_the_oracles_bits(hip):
    from gravit_amd import _build

    assert os.path.exists(_build.LIB_EXP), "libgvt_hip_exp.so is missing: __graft_entry__.build() makes it"
    env = dict(os.environ, GVT_HIP_LIB=_build.LIB_EXP)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "experiment_cases.py"), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    tail = p.stdout.decode(errors="replace")[-3000:]
    assert p.returncode == 0, tail
    assert " passed" in tail and "failed" not in tail, tail


@pytest.mark.gpu
def test_shipped_library_refuses_the_experiment_knobs(hip):
    from gravit_amd import capi

    assert capi.load().gvt_hip_is_experiments_build() == 0
    for k, v in (("trav_kernel", 0), ("wide4", 0), ("coop_fetch", 1), ("fused", 1), ("packet", 1), ("quad", 1)):
        with pytest.raises(capi.GvtHipError):
            hip.set_option(k, v)
    for k, v in (("trav_kernel", 1), ("wide4", 1), ("coop_fetch", 0), ("fused", 0), ("packet", 0), ("quad", 0)):
        hip.set_option(k, v)  # their shipped values are accepted
