"""The library's RCCL leg with MORE THAN ONE RANK, each rank a process of its own: what a multi-GPU run executes (ncclCommInitRank from a
broadcast unique id, one grouped ncclSend / ncclRecv exchange per tick with the payloads inline or behind, ncclReduce for the composite,
bench.py's self-started ranks) -- on the one GPU of this pool's boxes, where RCCL itself refuses two ranks on a device.  The transport
underneath is the stand-in tests/fake_rccl (same entry points, compiled against <rccl/rccl.h>, messages through files, every call
blocking; loaded through GVT_HIP_RCCL_LIB): it checks that every receive meets a send of exactly its size in issue order, which the real
library needs and never tells.  Says nothing about speed."""
import json
import os
import socket
import subprocess
import sys

import pytest

from tests.conftest import ROOT

pytestmark = pytest.mark.gpu
FAKE_DIR = os.path.join(ROOT, "tests", "fake_rccl")


def build_fake_rccl():
    so = os.path.join(FAKE_DIR, "libfakerccl.so")
    src = os.path.join(FAKE_DIR, "fake_rccl.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-fPIC", "-shared", "-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-o", so, src, "-L/opt/rocm/lib", "-lamdhip64"],
                       check=True, cwd=FAKE_DIR, timeout=300)
    return so


def rank_env(tmp_path, extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update({"GVT_HIP_RCCL_LIB": build_fake_rccl(), "TMPDIR": str(tmp_path), "FAKE_RCCL_TIMEOUT_S": "240", "HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    env.update(extra or {})
    return env


def free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


@pytest.mark.parametrize("world", [2, 3])
def test_domain_and_image_schedulers_across_processes(hip, tmp_path, world):
    """config 4, the config-5 slabs (payloads behind the announce, on the communicator's stream, nothing inline) and the soup tiles under the
    native Domain scheduler, the replicated Image scheduler, and the reference's own four CTest runs against its golden PPMs, on `world` processes: rank 0's composited image equals the checker's
    restated DomainTracer; rays sent / traced add up over the ranks to the checker's counts; a second frame of the same tracer agrees."""
    out = tmp_path / "verdict.json"
    env = rank_env(tmp_path, {"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(free_port()), "WORLD_SIZE": str(world)})
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "multiproc_worker.py"), str(out)], cwd=ROOT, env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=900)[0])
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
    assert all(p.returncode == 0 for p in procs), "\n---\n".join(lg[-3000:] for lg in logs)
    verdict = json.loads(out.read_text())
    assert verdict["world"] == world and len(verdict["cases"]) == 10
    assert all(c["ppm_sum_abs_diff"] < 300 for c in verdict["cases"] if "ppm_sum_abs_diff" in c)  # the reference's own distributed CTest, four runs
    by = {c["case"]: c for c in verdict["cases"]}
    assert by["config5 in 4 slabs, nothing inline"]["rays_inline"] == 0 and by["config4 asynchronous"]["rays_inline"] > 0
    assert all(c.get("rays_sent", 1) > 0 for c in verdict["cases"])
    assert not os.path.exists(os.path.join(str(tmp_path), "FAILED"))


def test_bench_script_starts_its_ranks_and_exchanges_over_the_rccl_leg(hip, tmp_path):
    """plain `python bench.py --gpus 2` (no launcher): the script starts two ranks, each builds the library's communicator and every variant
    and extra leg runs its exchanges through ncclSend / ncclRecv / ncclReduce (stand-in transport; --same-gpu: both ranks on device 0)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-gpu", "--tris", "200000", "--weak-tris", "100000", "--width", "480", "--height", "270",
                        "--config4-width", "380", "--config4-height", "216", "--steps", "2", "--warmup", "1"], cwd=ROOT, env=rank_env(tmp_path), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    j = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["config"]["harness"] == "native" and j["config"]["rccl_comm_count"] == 2
    assert set(j["variants"]) == {"domain_async", "domain_bsp", "image_replicated", "domain_async_known_miss_shortcut"}
    assert all("failed" not in v for v in j["variants"].values()), j["variants"]
    assert j["config"]["transport"] == "GVT_HIP_RCCL_LIB=libfakerccl.so" and "not a measurement" in j["config"]["rehearsal"]  # the line says what it is
    assert j["variants"]["domain_async"]["rays_sent_per_step"] > 0 and j["variants"]["domain_async"]["transport_groups_per_step"] > 0
    assert j["config4_bunny_grid"]["domain_async"]["value"] > 0 and j["weak_soup"]["tiles"] == 2 and j["weak_soup"]["value"] > 0
    # the line verifies itself: every variant's composited image bit for bit against rank 0's one-rank render, ray counts equal, deposit counts of the weak leg
    par = j["parity"]
    assert par["bit_exact"] is True and par["failed"] == [], par
    assert set(par["variants"]) == {"domain_async", "domain_bsp", "image_replicated", "domain_async_known_miss_shortcut", "config4_bunny_grid domain_async",
                                    "config4_bunny_grid domain_bsp", "weak_soup"}
    for name in ("domain_async", "domain_bsp", "image_replicated", "domain_async_known_miss_shortcut"):
        p_ = j["variants"][name]["parity"]
        assert p_["bit_exact"] and p_["rays_equal"] and p_["lit_pixels"] > 1000, (name, p_)
    assert "vs_strict_rule" in j["variants"]["domain_async_known_miss_shortcut"]["parity"]
    assert j["weak_soup"]["parity"]["deposits_equal"]
    assert par["one_rank_render_vs_cpu_oracle"]["bit_exact"] is True  # rank 0's own render of the un-cut soup against the CPU oracle's frame
    cb = j["cpu_baseline"]
    assert cb["value"] > 0 and cb["simd"]["value"] > 0 and "rank 0" in cb["note"] and j["roofline"]["frac"] > 0
