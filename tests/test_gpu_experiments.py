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
    for k, v in (("trav_kernel", 0), ("wide4", 0), ("coop_fetch", 1), ("fused", 1), ("quad", 1), ("refill_min", 8), ("inner_min", 16), ("share", 0),
                 ("blocks_per_cu", 5), ("wave_single", 0), ("shadow_direct", 0), ("lean_frame", 0), ("report_poll", 0), ("long_save", 0), ("top_ordered", 0), ("camera_tile", 0), ("abi_pipe_min", 0)):
        with pytest.raises(capi.GvtHipError):
            hip.set_option(k, v)
    for k, v in (("trav_kernel", 1), ("wide4", 1), ("coop_fetch", 0), ("fused", 0), ("quad", 0), ("refill_min", 16), ("inner_min", 32), ("share", 1)):
        hip.set_option(k, v)  # their shipped values are accepted
    # the shipped surface: 23 knobs (behaviour switches, budgets, test hooks)
    for k, v in (("shadow_order", 0), ("shadow_order_min_rays", 0), ("packet", 0), ("packet", 2), ("packet_sah_max", 64), ("packet_min_rays", 0), ("inline_kb", 0), ("comm_cus", 8), ("skip_known", 0), ("frame_timing", 1), ("term_sink", 0), ("sort_rays", 1), ("leaf_max", 4), ("long_steps", 64), ("long_min_rays", 0), ("long_auto", 0), ("finish_auto", 0), ("payload_overlap_kb", 0),
                 ("small_rays", 0), ("finish_rays", 0), ("round_room_mb", 0), ("abi_lanes", 0), ("abi_chunk", 65536), ("inject_fail_tick", -1)):
        hip.set_option(k, v)
    hip.set_option("defaults", 0)
    with pytest.raises(capi.GvtHipError):
        hip.set_option("no_such_knob", 1)
