"""world_size-2 (and 3) runs of the Domain scheduler's control flow over gloo on CPU: ray exchange in the
reference's 80-byte wire format, count all-gather, termination, framebuffer reduce.  The device work is done
by the checker backend (tests/oracle_backend.py); the scheduler code under test is the product's."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gravit_amd import scenes
from gravit_amd.scheduler import DomainTracer, ImageTracer
from tests.conftest import ROOT
from tests.helpers import oracle_render, oracle_render_domain
from tests.oracle_backend import OracleBackend


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(name):
    if name == "simple":
        sc = scenes.simple_scene(96, 96)
        owner = lambda world: [(sc.inst_mesh[i] % world) for i in range(sc.n_inst)]  # cones on rank 0, cubes on rank 1
    else:
        sc = scenes.soup_domains_scene(30000, 4, 96, 54)
        # look along -x so that rays cross the x-tiled domains one after the other; light off-axis
        sc.camera.eye, sc.camera.focus = (3.0, 0.6, 0.4), (0.5, 0.5, 0.5)
        sc.lights["position"] = (2.0, 2.5, 1.5)
        owner = lambda world: [i % world for i in range(sc.n_inst)]
    return sc, owner


def _worker(rank, world, port, name, outdir, overlap=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sc, owner_fn = _build(name)
        owner = owner_fn(world)
        owned = [o == rank for o in owner]
        tr = DomainTracer(sc, owner, dist, torch, "cpu", 1, backend=OracleBackend(sc, 1, owned), overlap=overlap)
        tr()
        fb = tr.composite()
        stats = torch.tensor([tr.rays_sent, tr.rounds, tr.adapter_calls], dtype=torch.int64)
        dist.all_reduce(stats)
        if rank == 0:
            np.save(os.path.join(outdir, "fb.npy"), fb)
            np.save(os.path.join(outdir, "stats.npy"), stats.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,world", [("simple", 2), ("soup", 2), ("soup", 3)])
def test_domain_tracer_over_gloo(tmp_path, name, world):
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path)), nprocs=world, join=True)
    fb = np.load(tmp_path / "fb.npy")
    stats = np.load(tmp_path / "stats.npy")
    sc, owner_fn = _build(name)
    owner = owner_fn(world)
    ref_fb, st = oracle_render_domain(sc, owner, world, 1)
    assert stats[0] > 0, "no rays crossed ranks: the exchange path was not exercised"
    assert stats[0] == st.rays_sent  # same rays cross the same rank boundaries as in the restated DomainTracer
    assert np.array_equal(fb[..., :3], ref_fb[..., :3])
    if name == "simple":  # and it is the 1-rank image (one writer per pixel)
        img_fb, _ = oracle_render(sc, 1)
        assert np.array_equal(fb[..., :3], img_fb[..., :3])


@pytest.mark.parametrize("name,world", [("simple", 2), ("soup", 3)])
def test_overlapped_domain_tracer_over_gloo(tmp_path, name, world):
    """overlap=True: exchanges posted before each local adapter call and completed after it, termination folded into the count
    exchange.  Same rays cross the same boundaries and the image is the BSP one."""
    mp.spawn(_worker, args=(world, _free_port(), name, str(tmp_path), True), nprocs=world, join=True)
    fb = np.load(tmp_path / "fb.npy")
    stats = np.load(tmp_path / "stats.npy")
    sc, owner_fn = _build(name)
    ref_fb, st = oracle_render_domain(sc, owner_fn(world), world, 1)
    assert stats[0] == st.rays_sent and stats[0] > 0
    assert np.array_equal(fb[..., :3], ref_fb[..., :3])


def test_image_tracer_with_checker_backend_matches_restated_loop():
    sc = scenes.simple_scene(64, 64)
    tr = ImageTracer(sc, 1, backend=OracleBackend(sc, 1))
    B = tr()
    ref_fb, st = oracle_render(sc, 1)
    assert tr.adapter_calls == st.adapter_calls
    assert np.array_equal(B.framebuffer(True)[..., :3], ref_fb[..., :3])


def test_single_rank_domain_tracer_needs_no_process_group():
    sc = scenes.simple_scene(48, 48)
    tr = DomainTracer(sc, [0] * sc.n_inst, dist, torch, "cpu", 1, backend=OracleBackend(sc, 1))
    tr()
    ref_fb, _ = oracle_render(sc, 1)
    assert np.array_equal(tr.composite()[..., :3], ref_fb[..., :3])


def test_bench_script_multi_rank_branch_under_gloo(tmp_path):
    """bench.py's N > 1 branch end to end (rendezvous, scene cut into one domain per rank, owner map, the timed loop, the reductions of
    the report, the JSON line) with --harness checker: world 2 over gloo on CPU, toy size.  A Python-level bug in that plumbing must not
    be what the first multi-GPU run finds; the native loop itself is covered on the device (tests/test_gpu_native.py)."""
    import json
    import subprocess
    import sys

    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--harness", "checker", "--tris", "20000",
                                       "--width", "96", "--height", "54", "--steps", "2", "--warmup", "1", "--weak-tris", "8000", "--config4-width", "95", "--config4-height", "54"],
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][1][-2000:] + outs[1][1][-2000:]
    line = [ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["scaling"] == "strong" and j["value"] > 0
    assert j["config"]["primary_traced_per_step"] > 1000 and j["config"]["shadow_traced_per_step"] > 100
    assert not [ln for ln in outs[1][0].splitlines() if ln.startswith("{")], "only rank 0 prints the line"
    # the scheduler variants measured in the same invocation (the native harness adds the replicated Image scheduler)
    v = j["variants"]
    assert set(v) == {"domain_async", "domain_bsp"} and v["domain_async"]["is_value"] and not v["domain_bsp"]["is_value"]
    for r in v.values():
        assert r["value"] > 0 and r["ms_per_step"] > 0 and set(r["phase_ms_per_step_max_over_ranks"]) == {"chain", "announce", "payload", "composite", "host_wait"}
    # the two extra legs: BASELINE configs[3] (the 8-bunny grid, domain d on rank d mod N) and the weak-scaling soup (a tile per rank)
    c4, wk = j["config4_bunny_grid"], j["weak_soup"]
    for r in (c4["domain_async"], c4["domain_bsp"], wk):
        assert r["value"] > 0 and r["rays_per_step"] > 100 and r["ticks_per_step"] >= 1
    assert c4["film"] == [95, 54] and c4["domain_async"]["rays_sent_per_step"] == c4["domain_bsp"]["rays_sent_per_step"] > 0
    assert wk["scaling"] == "weak" and wk["tiles"] == 2 and wk["tris_per_tile"] == 8000 and wk["film"] == [136, 80] and len(wk["roofline_per_rank"]) == 2
    # the line verifies itself (the reference never runs distributed without diffing the image, CMakeLists.txt:650-688): every variant's composited image
    # against rank 0's one-rank render of the same scene, the rays traced against the one-rank counts, the weak leg's deposit counts; and it carries the CPU column
    par = j["parity"]
    assert par["bit_exact"] is True and par["failed"] == [] and par["skipped"] is None
    assert set(par["variants"]) == {"domain_async", "domain_bsp", "config4_bunny_grid domain_async", "config4_bunny_grid domain_bsp", "weak_soup"}
    for name, r in (("domain_async", v["domain_async"]), ("domain_bsp", v["domain_bsp"]), ("c4a", c4["domain_async"]), ("c4b", c4["domain_bsp"])):
        p_ = r["parity"]
        assert p_["bit_exact"] and p_["rays_equal"] and p_["max_abs_diff"] == 0.0 and p_["lit_pixels"] == p_["lit_pixels_got"] > 100, (name, p_)
        assert p_["rays_closest"][0] == p_["rays_closest"][1] > 0 and p_["rays_any"][0] == p_["rays_any"][1] > 0
    assert wk["parity"]["deposits_equal"] and wk["parity"]["deposits_composited"] == wk["parity"]["deposits_summed_over_ranks"] > 0
    cb = j["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and "rank 0" in cb["note"]


def test_bench_script_fails_loudly_when_a_variant_image_differs(tmp_path):
    """One damaged pixel in one variant's composited image (test hook GVT_BENCH_BREAK_PARITY): the line is still printed, says which variant failed, and
    the run leaves with a non-zero status -- through the script's own launcher too."""
    import json
    import subprocess
    import sys

    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--harness", "checker", "--tris", "20000", "--width", "96", "--height", "54",
                        "--steps", "1", "--warmup", "0", "--no-extra-legs", "--no-cpu-baseline"], env=dict(_clean_env(), GVT_BENCH_BREAK_PARITY="domain_bsp"), capture_output=True, text=True, timeout=300)
    assert p.returncode == 4, p.stderr[-3000:]
    j = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["value"] > 0 and j["parity"]["bit_exact"] is False and j["parity"]["failed"] == ["domain_bsp"]
    assert j["variants"]["domain_bsp"]["parity"]["pixels_differ"] == 1 and j["variants"]["domain_async"]["parity"]["bit_exact"]
    assert "parity FAILED for domain_bsp" in p.stderr


def test_bench_script_keeps_its_line_when_a_secondary_leg_fails(tmp_path):
    """A scheduler variant that fails in the ray exchange (here: injected on every rank, as a peer's error word or a passed deadline does it)
    must not cost the measurement made before it: the line is printed with `value`, the failed variant and `legs_error`; the legs behind it
    (same communicator) are skipped; every rank leaves with status 0 so that the launcher does not stop rank 0 before its line is out."""
    import json
    import subprocess
    import sys

    port = _free_port()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", OMP_NUM_THREADS="1", GVT_BENCH_FAIL_LEG="domain_bsp")
    procs = []
    for r in range(2):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--harness", "checker", "--tris", "20000",
                                       "--width", "96", "--height", "54", "--steps", "1", "--warmup", "0", "--weak-tris", "8000", "--config4-width", "95", "--config4-height", "54"],
                                      env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs[0][1][-2000:] + outs[1][1][-2000:]
    j = json.loads([ln for ln in outs[0][0].splitlines() if ln.startswith("{")][-1])
    assert j["value"] > 0 and "domain_bsp" in j["legs_error"]
    assert j["variants"]["domain_async"]["value"] > 0 and "injected" in j["variants"]["domain_bsp"]["failed"]
    assert "config4_bunny_grid" not in j and "weak_soup" not in j
    assert "domain_bsp failed" in outs[1][1]  # every rank says what it knows on stderr


def _clean_env():
    """the environment of a caller that knows nothing of the launcher: no RANK / WORLD_SIZE / MASTER_*"""
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    e["OMP_NUM_THREADS"] = "1"
    return e


def test_bench_script_starts_its_own_ranks_when_no_launcher_did(tmp_path):
    """`python bench.py --gpus 2` with NO launcher and no WORLD_SIZE in the environment (how the driver runs the N = 1 line): the script
    starts its two ranks itself, rank 0's single JSON line comes out of the parent's standard output, the status is 0."""
    import json
    import subprocess
    import sys

    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--harness", "checker", "--tris", "20000", "--width", "96", "--height", "54",
                        "--steps", "1", "--warmup", "0", "--no-extra-legs"], env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line, rank 0's"
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and set(j["variants"]) == {"domain_async", "domain_bsp"}


def test_bench_script_launcher_reports_a_failing_rank(tmp_path):
    """a rank that dies takes the launcher's status with it (non-zero), the surviving ranks are ended by their PIDs after the grace period"""
    import subprocess
    import sys

    env = dict(_clean_env(), GVT_BENCH_DIE_RANK="1", GVT_BENCH_SPAWN_GRACE_S="3")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--harness", "checker", "--tris", "20000", "--width", "96", "--height", "54",
                        "--steps", "1", "--warmup", "0", "--no-extra-legs"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "a rank left with status" in p.stderr
    assert not [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
