#!/usr/bin/env python3
"""bench.py -- Mrays/s (primary + shadow) of the GraviT adapter hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 10,000,000 random triangles,
1920x1080, one point light at the eye, depth 1 (primary + 1 shadow ray), AO off.  A step = one frame:
camera rays -> top-level test -> Adapter::trace (closest hit, shade, shadow rays, any hit) -> shuffle ->
framebuffer, all resident in HBM, driven by the native scheduler loop (gvt_hip_tracer_frame).  N = 1: Image
scheduler, one domain.  N > 1: the same soup cut into N spatial domains, one per GPU, Domain scheduler with the
ray exchange as RCCL point-to-point issued by the library itself (strong scaling: the frame is fixed);
torch.distributed (gloo) only carries the rendezvous: the RCCL unique id, barriers, the sums of the report.
value = (rays through the closest-hit kernel + rays through the any-hit kernel, all ranks) / wall time of the
timed steps.

  python bench.py --gpus 1 --steps 10 --warmup 2
  python bench.py --gpus N ...        (no launcher: the script starts its N ranks itself, spawn_ranks)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import math
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def algorithmic_bytes_per_ray(n_tris):
    """32 (o,d,tmin,tmax) + 16 (t,prim,u,v) + 32*ceil(log2(T/4)) (one box per level) + 4*48 (one leaf)."""
    levels = max(1, math.ceil(math.log2(max(n_tris, 8) / 4.0)))
    return 32 + 16 + 32 * levels + 4 * 48


def host_cores():
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (a GPU box hands a 1-GPU job a share)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = max(1, min(n, int(math.ceil(int(quota) / int(period)))))
    except Exception:
        pass
    return n


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True, timeout=5).stdout.strip() or None
    except Exception:
        return None


def source_hash():
    try:
        from gravit_amd import _build

        return _build.source_hash()
    except Exception:
        return None


def cpu_simd(adapter, om, s_all, moved, nthreads, repeats, which=4):
    """The SIMD CPU baseline (oracle/simd_baseline.c: one ray against four quantised child boxes per step in SSE4.1 + FMA, the GPU-built 4-wide tree
    downloaded once -- the closest stand-in for the reference's Embree BVH4 traversal that builds here) on the frame's primary rays (closest hit) and the
    shadow rays the oracle generated for them (any hit); its hits are checked against the oracle's before its time counts."""
    import numpy as np

    from oracle import simd

    t0 = time.perf_counter()
    nodes4, slots = adapter.download_wide()
    T = simd.Tree(nodes4, slots)
    dl = time.perf_counter() - t0
    wide8 = None
    if which == 8:
        if simd.load8() is None:
            return {"value": None, "skipped": "the host has no AVX2 + FMA (or oracle/libsimd8_baseline.so is missing)"}
        t0 = time.perf_counter()
        T = wide8 = simd.Tree8(nodes4, slots)
        dl += time.perf_counter() - t0
    po, pd = np.ascontiguousarray(s_all["origin"]), np.ascontiguousarray(s_all["direction"])
    sh = moved[moved["type"] == 1]
    so, sd = np.ascontiguousarray(sh["origin"]), np.ascontiguousarray(sh["direction"])
    best = None
    for _ in range(repeats):
        t0 = time.perf_counter()
        hits = T.intersect(po, pd, nthreads)
        steps_c = T.last_steps
        occ = T.occluded(so, sd, nthreads)
        d = time.perf_counter() - t0
        best = d if best is None else min(best, d)
    k = max(1, len(po) // 20000)  # validation: every k-th primary ray against the oracle's closest hit, bit for bit; the oracle forwarded every shadow ray of `moved`: none occluded
    ref = om.intersect(po[::k], pd[::k])
    same = bool(np.array_equal(ref["t"], hits[0][::k]) and np.array_equal(ref["prim"], hits[1][::k]) and np.array_equal(ref["u"], hits[2][::k]) and np.array_equal(ref["v"], hits[3][::k]))
    ok = same and not bool(occ.any())
    sel = slice(None, None, 16)
    t0 = time.perf_counter()
    T.intersect(po[sel], pd[sel], 1); T.occluded(so[sel], sd[sel], 1)
    d1 = time.perf_counter() - t0
    n1 = len(po[sel]) + len(so[sel])
    return {"value": (len(po) + len(so)) / best / 1e6 if ok else None, "unit": "Mrays/s", "cores": nthreads, "value_1_thread": n1 / d1 / 1e6 if ok else None,
            "validated_against_oracle": {"primary_rays_compared_bit_for_bit": int(len(ref)), "equal": same, "shadow_rays_all_unoccluded_like_the_oracle": not bool(occ.any())},
            "node_steps_per_primary_ray": steps_c[0] / max(1, len(po)), "leaf_steps_per_primary_ray": steps_c[1] / max(1, len(po)),
            "sample": "%d primary rays (closest hit) + %d shadow rays (any hit) of the frame in %.4f s wall on %d threads (best of %d); 1 thread: every 16th ray, %d rays in %.3f s; "
                      "tree download %.2f s excluded (like the build)" % (len(po), len(so), best, nthreads, repeats, n1, d1, dl),
            "what": ("one ray vs four quantised child boxes per step, SSE4.1 + FMA, nearest child first, chunks of 4096 rays over pthreads; the GPU-built compressed 4-wide tree "
                     "(gvt_hip_mesh_download_wide); stand-in for Embree 2.x's BVH4 single-ray traversal (EmbreeMeshAdapter.cpp:474, :375), which is not in the tree") if wide8 is None else
                    ("one ray vs EIGHT child boxes per step, AVX2 + FMA (oracle/simd8_baseline.c), nearest child first, chunks of 4096 rays over pthreads; the GPU-built 4-wide tree collapsed "
                     "once more on the host into %d 8-wide nodes of float boxes (%.2f children each); stand-in for Embree 2.x's BVH8 single-ray traversal on an AVX2 host -- the "
                     "reference picks its width from the host ISA (EmbreeMeshAdapter.cpp:50-74)" % (wide8.n8, wide8.children_per_node))}


def cpu_baseline(scene, row_stride, nthreads, repeats=3, gpu_fb=None, parity_out=None, adapter=None):
    """The CPU side of the report, timed on a bounded sample of the same frame (every row_stride-th scanline of the 1080p camera; default: the whole
    frame) on all usable host cores and on ONE thread: (1) the oracle itself (`scalar_port`: a scalar port of the reference's Embree adapter path over a
    median-split BVH2) and (2) the SIMD baseline (cpu_simd); `value` is the faster of the two -- Embree 2.x itself is not in the tree."""
    import numpy as np

    from oracle import orc

    cam = scene.camera
    m = scene.meshes[0]
    om = orc.Mesh(m.verts, m.tris, mesh_mat=m.material)  # BVH build excluded, like on the GPU side
    rays = orc.camera_rays(cam.eye, cam.focus, cam.up, cam.fov, cam.width, cam.height)

    def sample(stride):
        rows = np.arange(0, cam.height, stride)
        sel = (rows[:, None] * cam.width + np.arange(cam.width)[None, :]).reshape(-1)
        s = rays[sel].copy()
        nxt, tt = orc.toplevel_intersect(scene.inst_lo, scene.inst_hi, [0], s)
        hit = nxt >= 0
        s2 = s[hit].copy()
        s2["origin"] += s2["direction"] * (tt[hit] * np.float32(0.95))[:, None]
        return s2

    s_all = sample(row_stride)
    dt, moved = None, None
    for _ in range(repeats):
        rays_in = s_all.copy()
        t0 = time.perf_counter()
        moved = om.trace(rays_in, scene.m[0], scene.minv[0], scene.normi[0], scene.lights, 0, 0, nthreads)
        d = time.perf_counter() - t0
        dt = d if dt is None else min(dt, d)
    c, a = orc.trace_counts()
    if gpu_fb is not None and row_stride == 1 and scene.n_inst == 1:
        # the checker's frame (one domain: every un-occluded shadow ray ends in the framebuffer, TracerBase.h:396-400) against the
        # frame the timed loop left in HBM -- the whole 1080p image at the benchmark's full size
        ref = np.zeros((cam.height * cam.width, 4), np.float32)
        sh = moved[(moved["type"] == 1)]
        np.add.at(ref[:, :3], sh["id"], sh["color"] * sh["w"][:, None])
        np.add.at(ref[:, 3], sh["id"], 1.0)
        got = np.asarray(gpu_fb, np.float32).reshape(-1, 4)
        parity_out["parity"] = {"checked": "whole %dx%d frame of the timed loop vs the CPU oracle" % (cam.width, cam.height),
                                "lit_pixels": int((ref[:, 3] > 0).sum()), "max_abs_diff": float(np.abs(got - ref).max()),
                                "bit_exact": bool(np.array_equal(got, ref))}
    s_one = sample(16 * row_stride)
    t0 = time.perf_counter()
    om.trace(s_one.copy(), scene.m[0], scene.minv[0], scene.normi[0], scene.lights, 0, 0, 1)
    dt1 = time.perf_counter() - t0
    c1, a1 = orc.trace_counts()
    scalar = {"value": (c + a) / dt / 1e6, "unit": "Mrays/s", "cores": nthreads, "value_1_thread": (c1 + a1) / dt1 / 1e6,
              "sample": "scanlines 0,%d,.. of the %dx%d frame: %d primary + %d shadow rays in %.3f s wall on %d pinned threads (best of %d); "
                        "1 thread: scanlines 0,%d,..: %d rays in %.3f s; BVH build excluded; the CPU oracle = scalar port of the Embree adapter path over a median-split BVH2 "
                        "(whole trace: traversal, shading, shadow-ray generation)" % (row_stride, cam.width, cam.height, c, a, dt, nthreads, repeats, 16 * row_stride, c1 + a1, dt1)}
    out = dict(scalar, kind="port", cpu_model=cpu_model(), which="scalar_port", scalar_port=scalar)
    if adapter is not None:
        for key, width in (("simd", 4), ("simd8", 8)):  # SSE one ray x 4 boxes; AVX2 one ray x 8 boxes
            try:
                sd = cpu_simd(adapter, om, s_all, moved, nthreads, repeats, which=width)
                out[key] = sd
                if sd.get("value") is not None and sd["value"] > out["value"]:  # `value`: the best CPU figure
                    out.update({"value": sd["value"], "value_1_thread": sd["value_1_thread"], "which": key, "sample": sd["sample"]})
            except Exception as e:  # noqa: BLE001
                out[key] = {"value": None, "failed": repr(e)}
    return out


def measured_stream_peak(torch, dev):
    """HBM rate of a plain device-to-device copy of 1 GiB (read + write), best of 5: the achievable peak beside the 8 TB/s spec."""
    n = 1 << 28
    a = torch.empty(n, dtype=torch.float32, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    torch.cuda.synchronize()
    best = 0.0
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        best = max(best, 2.0 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a, b
    return best


def abi_path(scene, tracer, capi, np):
    """The literal drop-in path: Adapter::trace on a HOST RayVector (gvt_hip_trace: 80-byte rays in over PCIe, moved rays + the
    updated rayList back), timed on the frame's primary rays."""
    B = tracer.backend
    B.begin_frame()
    B.generate_and_filter(None)  # the frame's primary rays as the scheduler hands them to the adapter (device camera + FilterRaysLocally)
    s2 = B.queues[0].to_numpy()
    B.begin_frame()
    ad = B.adapter(0)
    from gravit_amd.layouts import RAY_DTYPE

    moved = np.zeros(len(s2) * (1 + len(scene.lights)), RAY_DTYPE)
    moved[:] = moved  # touched once: a scheduler re-uses its moved_rays vector, first-touch page faults are not part of a call
    res = {}
    for key, wb in (("write_back", True), ("no_write_back", False)):
        best, n_out = None, 0
        for _ in range(4):
            r = s2.copy()
            t0 = time.perf_counter()
            out = ad.trace(r, scene.m[0], scene.minv[0], scene.normi[0], scene.lights, write_back=wb, out=moved)
            d = time.perf_counter() - t0
            best = d if best is None else min(best, d)
            n_out = len(out)
        n_traced = len(s2) + n_out  # depth 1, light at the eye: every shadow ray survives (31 of 1 M are occluded)
        bytes_pcie = 80 * ((2 if wb else 1) * len(s2) + n_out)
        res[key] = {"Mrays/s": n_traced / best / 1e6, "ms": best * 1e3, "pcie_GB/s": bytes_pcie / best / 1e9}
    out = dict(res["write_back"])
    out.update({"rays_in": int(len(s2)), "rays_out": int(n_out), "no_write_back": res["no_write_back"],
                "note": "gvt_hip_trace on host rays (pageable numpy buffers, moved_rays buffer re-used), best of 4: H2D of the 80-byte rays, "
                        "trace, D2H of the updated rayList and of moved_rays; no_write_back = GVT_HIP_TRACE_NO_WRITEBACK (the reference's "
                        "schedulers clear the traced queue right after the call); never `value`"})
    return out


def sustained(frame, rays_per_frame, seconds=2.5, min_frames=2000):
    """The same frame rendered back to back for at least `seconds` and `min_frames` frames (the headline is a window of K = 20 frames
    behind the warm-up): does the rate survive seconds of load (clocks, thermals)?  Never `value`."""
    import numpy as np

    ts = []
    t0 = time.perf_counter()
    while len(ts) < min_frames or time.perf_counter() - t0 < seconds:
        f0 = time.perf_counter()
        frame()
        ts.append(time.perf_counter() - f0)
        if len(ts) >= 200_000:
            break
    el = time.perf_counter() - t0
    a = np.array(ts) * 1e3
    k = max(1, len(a) // 10)
    return {"frames": len(ts), "seconds": el, "mean_ms": float(a.mean()), "p50_ms": float(np.percentile(a, 50)), "p99_ms": float(np.percentile(a, 99)),
            "max_ms": float(a.max()), "first_tenth_mean_ms": float(a[:k].mean()), "last_tenth_mean_ms": float(a[-k:].mean()),
            "Mrays/s": rays_per_frame * len(ts) / el / 1e6,
            "note": "frames rendered back to back, one in flight, each ending with its host synchronisation; wall clock over the whole run; never `value`"}


def two_frames_in_flight(scene, steps, rays_per_frame):
    """Throughput of a STREAM of frames: two tracers, each with its own library context (stream, scratch, counters, queues,
    framebuffer), render alternate frames from two host threads; one frame's launch tails overlap the other's bulk.  Reported
    beside the line, never as `value` (which keeps one frame in flight, like the reference's frame loop)."""
    import threading

    from gravit_amd.layouts import NORMALS_FLAT
    from gravit_amd.scheduler import Context, NativeTracer

    ready, go = threading.Barrier(3), threading.Barrier(3)
    spans, errs = [None, None], []

    def work(k):
        try:
            with Context(0):
                tr = NativeTracer(scene, NORMALS_FLAT)
                for _ in range(3):
                    tr()
                ready.wait(); go.wait()
                t0 = time.perf_counter()
                for _ in range(steps):
                    tr()
                spans[k] = (t0, time.perf_counter())
                tr.close()
                tr = None
        except Exception:  # noqa: BLE001
            import traceback
            errs.append(traceback.format_exc())
            for b in (ready, go):
                b.abort()

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    try:
        ready.wait(); go.wait()
    except threading.BrokenBarrierError:
        pass
    [t.join() for t in th]
    if errs:
        raise RuntimeError(errs[0])
    span = max(e for _, e in spans) - min(b for b, _ in spans)
    return {"ms_per_frame": span / (2 * steps) * 1e3, "Mrays/s": rays_per_frame * 2 * steps / span / 1e6, "frames": 2 * steps,
            "note": "two library contexts on two host threads, each rendering %d frames of the same workload concurrently; first start "
                    "to last end; never `value`" % steps}


def one_rank_reference(sc, mode, harness, on_gpu, capi, skip_known=0, skip_known_restore=0):
    """Rank 0 of an N > 1 run: the SAME scene rendered by ONE rank -- every domain local, no communicator, no exchange -- on this rank's own device,
    as the thing every multi-rank variant's composited framebuffer must equal (the reference never runs distributed without diffing the image
    afterwards: CMakeLists.txt:650-654 `ibrun -np 2 ...`, :664-682 gvtImageDiff).  Returns (float framebuffer un-clamped, rays through the
    closest-hit kernels, rays through the any-hit kernels) of one frame."""
    if harness == "native":
        from gravit_amd.scheduler import NativeTracer

        capi.set_option("skip_known", int(skip_known))
        try:
            tr = NativeTracer(sc, mode)
            tr()
            fb, st = tr.backend.framebuffer(False), dict(tr.stats)
            tr.close()
        finally:
            capi.set_option("skip_known", int(skip_known_restore))
        return fb, int(st["rays_closest"]), int(st["rays_any"])
    from gravit_amd.scheduler import ImageTracer  # the Python harness: the reference-order loop over one backend that holds every instance

    if on_gpu:
        t = ImageTracer(sc, mode)
        c0 = capi.stats()
        t()
        c1 = capi.stats()
        return t.backend.framebuffer(False), int(c1["rays_closest"] - c0["rays_closest"]), int(c1["rays_any"] - c0["rays_any"])
    from tests.oracle_backend import OracleBackend

    be = OracleBackend(sc, mode, None)
    ImageTracer(sc, mode, backend=be)()
    return be.framebuffer(False), int(be.rays_closest), int(be.rays_any)


def frame_parity(name, got, ref, rays_got, rays_ref):
    """One variant's composited framebuffer (rank 0, un-clamped float sums + the deposit count in alpha) against the one-rank render of the same
    scene, and the rays traced per frame summed over the ranks against the one-rank counts (the strict shuffle rule moves the same rays through the same
    instances however they are spread over ranks)."""
    import numpy as np

    got = np.asarray(got, np.float32).reshape(-1, 4)
    ref = np.asarray(ref, np.float32).reshape(-1, 4)
    if os.environ.get("GVT_BENCH_BREAK_PARITY") == name:  # test hook (tests/test_domain_gloo.py): one pixel of this variant's image is damaged before the comparison
        got = got.copy()
        got[int(np.argmax(ref[:, 3] > 0)), 0] += 0.25
    same = bool(np.array_equal(got, ref))
    return {"checked": "rank 0's composited framebuffer vs the same scene rendered by ONE rank (all domains local, no exchange) on rank 0's device",
            "bit_exact": same, "lit_pixels": int((ref[:, 3] > 0).sum()), "lit_pixels_got": int((got[:, 3] > 0).sum()),
            "deposits": float(ref[:, 3].astype(np.float64).sum()), "deposits_got": float(got[:, 3].astype(np.float64).sum()),
            "max_abs_diff": float(np.abs(got - ref).max()) if got.size else 0.0, "pixels_differ": int((got != ref).any(axis=1).sum()),
            "rays_closest": [int(round(rays_got[0])), int(rays_ref[0])], "rays_any": [int(round(rays_got[1])), int(rays_ref[1])],
            "rays_equal": bool(int(round(rays_got[0])) == int(rays_ref[0]) and int(round(rays_got[1])) == int(rays_ref[1]))}


PHASES = ("ms_chain", "ms_announce", "ms_payload", "ms_composite", "ms_host_wait")


def die_on_exchange_error(rank, world, what, err, stats):
    """A frame failed under several ranks (a deadline passed, a peer reported an error): say what this rank knows and leave with a
    non-zero status -- a fresh launch is the launcher's business, this process never re-executes itself."""
    print("bench.py: rank %d of %d: %s failed: %s\n  last frame stats of this rank: %s" % (rank, world, what, err, stats), file=sys.stderr, flush=True)
    os._exit(1)


class LegFailed(Exception):
    """A frame of a secondary measurement (a scheduler variant, an extra leg) failed in the ray exchange.  Such a failure is collective -- the
    failing rank's error word travels in the announce, a silent peer runs everyone into the deadline -- so every rank gets here at the
    same exchange: the leg is recorded as failed, the legs behind it (same communicator) are skipped, `value` (measured before) stands."""


def measure_variant(run_frame, frame_stats, steps, warmup, barrier, reduce_sum, reduce_max, rank, world, name):
    """W untimed + K timed frames of one scheduler variant; rays and bytes summed over the ranks, wall time and the per-phase times
    as the maximum over the ranks.  run_frame() renders one frame, frame_stats() returns that frame's counters on this rank."""
    def frame():
        try:
            if os.environ.get("GVT_BENCH_FAIL_LEG") == name:  # test hook (tests/test_domain_gloo.py): this leg fails on every rank, as an exchange failure does
                raise type("GvtHipError", (Exception,), {})("injected failure (GVT_BENCH_FAIL_LEG)")
            run_frame()
        except Exception as e:  # noqa: BLE001 -- GvtHipError: a deadline passed or a peer reported an error
            if type(e).__name__ != "GvtHipError":
                raise
            print("bench.py: rank %d of %d: %s failed: %s\n  last frame stats of this rank: %s" % (rank, world, name, e, frame_stats()), file=sys.stderr, flush=True)
            raise LegFailed("%s: %s" % (name, e)) from e

    barrier()  # the ranks prepared their scenes at their own pace: start together, inside the exchange's deadline
    for _ in range(warmup):
        frame()
    frame_stats()  # (a harness that counts through running totals hands out differences: what the warm-up traced is not the timed steps')
    barrier()
    sums = {}
    t0 = time.perf_counter()
    for _ in range(steps):
        frame()
        for k, v in frame_stats().items():
            sums[k] = sums.get(k, 0) + v
    barrier()
    elapsed = time.perf_counter() - t0
    tot = reduce_sum([float(sums.get(k, 0)) for k in ("rays_closest", "rays_any", "rays_sent", "bytes_sent", "rays_inline")])
    mx = reduce_max([elapsed] + [float(sums.get(k, 0.0)) for k in PHASES] + [float(sums.get(k, 0)) for k in ("rounds", "chains", "host_syncs", "exchanges")])
    ticks = mx[1 + len(PHASES)] / steps
    res = {"value": (tot[0] + tot[1]) / mx[0] / 1e6, "unit": "Mrays/s", "ms_per_step": mx[0] / steps * 1e3,
           "ticks_per_step": ticks, "ms_per_tick": (mx[0] / steps * 1e3 / ticks) if ticks else None,
           "launch_chains_per_step": mx[2 + len(PHASES)] / steps, "host_syncs_per_step": mx[3 + len(PHASES)] / steps,
           "rays_sent_per_step": tot[2] / steps, "bytes_sent_per_step": tot[3] / steps,
           "rays_sent_inline_per_step": tot[4] / steps,  # of rays_sent: rays that travelled inside an announce (one transport group per tick)
           "transport_groups_per_step": mx[4 + len(PHASES)] / steps,  # ncclGroupStart .. End per rank and frame (max over ranks), composite included
           "phase_ms_per_step_max_over_ranks": {k[3:]: mx[1 + i] / steps for i, k in enumerate(PHASES)}}
    return res, sums, mx[0], tot


def weak_film(n, width, height):
    """weak_soup: the film grows with the rank count (area x N, 16:9 kept, multiples of 8), so that every rank keeps about the one-GPU
    benchmark's number of rays as well as its number of triangles."""
    s = math.sqrt(n)
    return max(8, int(round(width * s / 8.0)) * 8), max(8, int(round(height * s / 8.0)) * 8)


def rank_roofline(st, rays_closest, rays_any, n_tris_local):
    """This rank's dominant traversal kernel against the HBM roofline (algorithmic bytes) from the library's per-kernel HIP events."""
    dom = "closest" if st["ms_closest"] >= st["ms_any"] else "any"
    ms, nl = st["ms_%s" % dom], max(1, st["launches_%s" % dom])
    rays = rays_closest if dom == "closest" else rays_any
    b_ray = algorithmic_bytes_per_ray(n_tris_local)
    ach = (rays * b_ray / nl) / (ms / nl / 1e3) / 1e9 if ms > 0 else 0.0
    return {"bound": "hbm", "kernel": dom, "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_ray": b_ray, "rays_per_launch": rays / nl, "avg_launch_ms": ms / nl, "launches": nl}


CONFIG4_NOTE = ("BASELINE configs[3]: 8 bunny instances on a 4x2 grid (pitch 0.3, data/bunny.conf:27-29) at the film of data/bunny.conf:8, smooth normals, "
                "Domain scheduler, domain d on rank d mod N (DomainTracer.h:115-144); never `value`")
WEAK_NOTE = ("weak scaling of config 3: N tiles of %d triangles EACH (every rank generates and holds only its own tile; triangle size ~ N^(-1/3): config 3's "
             "extent-to-spacing ratio), film %dx%d (area x N), Domain scheduler asynchronous; ideal = N x the one-GPU value; never `value`")


def run_inproc(scene, owner, N, mode, call, replicate, args, roofline=False):
    """One scheduler variant on N in-process ranks (threads of this process, one context each, the library's in-process transport)."""
    import threading

    from gravit_amd import capi
    from gravit_amd.scheduler import Comm, Context, NativeTracer

    hub = capi.load().gvt_hip_hub_create(N)
    bar = threading.Barrier(N)
    per_rank, errs = {}, []

    def rank_main(r):
        ctx = None
        try:
            ctx = Context(0)
            capi.set_option("frame_timing", 1)  # the per-phase breakdown (five more event calls per exchange: ms_per_step here includes them)
            for o in args.opt:
                k, v = o.split("=")
                capi.set_option(k, int(v))
            comm = Comm.local(hub, r)
            tr = NativeTracer(scene, mode, owner, comm, replicate=replicate)
            for _ in range(args.warmup):
                tr(**call)
            capi.synchronize(); bar.wait()
            if roofline:
                capi.stats_reset(); capi.profile(2)
            sums = {}
            t0 = time.perf_counter()
            for _ in range(args.steps):
                tr(**call)
                for k, v in tr.stats.items():
                    sums[k] = sums.get(k, 0) + v
            capi.synchronize(); bar.wait()
            el = time.perf_counter() - t0
            roof = None
            if roofline:
                st = capi.stats(); capi.profile(False)
                mine = [scene.meshes[scene.inst_mesh[i]] for i in range(scene.n_inst) if owner[i] == r and scene.meshes[scene.inst_mesh[i]] is not None]
                roof = rank_roofline(st, sums.get("rays_closest", 0), sums.get("rays_any", 0), max([len(m.tris) for m in mine] or [8]))
            per_rank[r] = (sums, el, roof)
            tr.close(); comm.close()
            tr = None
        except Exception:  # noqa: BLE001
            import traceback
            errs.append(traceback.format_exc())
            capi.load().gvt_hip_hub_abort(hub)
            bar.abort()
        finally:
            import gc
            gc.collect()
            if ctx is not None:
                ctx.close()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(N)]
    [t.start() for t in th]
    [t.join() for t in th]
    capi.load().gvt_hip_hub_destroy(hub)
    if errs:
        print(errs[0], file=sys.stderr)
        sys.exit(1)
    el = max(v[1] for v in per_rank.values())
    tot = lambda k: float(sum(v[0].get(k, 0) for v in per_rank.values()))  # noqa: E731
    mx = lambda k: float(max(v[0].get(k, 0) for v in per_rank.values()))  # noqa: E731
    ticks = mx("rounds") / args.steps
    res = {"value": (tot("rays_closest") + tot("rays_any")) / el / 1e6, "unit": "Mrays/s", "ms_per_step": el / args.steps * 1e3,
           "ticks_per_step": ticks, "ms_per_tick": el / args.steps * 1e3 / ticks if ticks else None,
           "launch_chains_per_step": mx("chains") / args.steps, "host_syncs_per_step": mx("host_syncs") / args.steps,
           "rays_per_step": (tot("rays_closest") + tot("rays_any")) / args.steps,
           "rays_sent_per_step": tot("rays_sent") / args.steps, "bytes_sent_per_step": tot("bytes_sent") / args.steps,
           "rays_sent_inline_per_step": tot("rays_inline") / args.steps, "transport_groups_per_step": mx("exchanges") / args.steps,
           "phase_ms_per_step_max_over_ranks": {k[3:]: mx(k) / args.steps for k in PHASES}}
    if roofline:
        res["roofline_per_rank"] = [per_rank[r][2] for r in range(N)]
    return res


def inproc_ranks(args):
    """--inproc-ranks N: the native multi-rank frame loop (announces, wire packing, payload, vote, composite) with the N ranks as
    threads of THIS process on ONE GPU, joined by the library's in-process transport: the same control flow and the same keys as a
    --gpus N run, for looking at tick counts and per-tick cost where only one GPU is at hand.  Not a scaling number."""
    from gravit_amd import capi, scenes
    from gravit_amd.layouts import NORMALS_FLAT, NORMALS_SMOOTH

    N = args.inproc_ranks
    capi.init(0)
    variants = [("domain_async", "domain", dict(bsp=False)), ("domain_bsp", "domain", dict(bsp=True)), ("image_replicated", "image", dict(image=True))]
    scene_dom = scenes.soup_domains_scene(args.tris, N, args.width, args.height)
    scene_img = scenes.soup_scene(args.tris, args.width, args.height)
    out = {"metric": "Mrays/s (primary+shadow) at 1080p, 10M-tri scene", "n_gpus": 1, "inproc_ranks": N, "steps": args.steps, "warmup": args.warmup,
           "note": "N in-process ranks share ONE GPU (hub transport): tick counts, bytes and per-phase times of the Domain scheduler's frame loop; not a scaling number",
           "variants": {}}
    for name, kind, call in variants:
        scene = scene_dom if kind == "domain" else scene_img
        owner = [i % N for i in range(scene.n_inst)] if kind == "domain" else [0] * scene.n_inst
        out["variants"][name] = run_inproc(scene, owner, N, NORMALS_FLAT, call, kind == "image", args)
    scene_dom = scene_img = None
    if not args.no_extra_legs:
        sc4 = scenes.bunny_grid_scene(width=args.config4_width, height=args.config4_height)
        own4 = [i % N for i in range(sc4.n_inst)]
        out["config4_bunny_grid"] = {"workload": CONFIG4_NOTE, "film": [args.config4_width, args.config4_height],
                                     "domain_async": run_inproc(sc4, own4, N, NORMALS_SMOOTH, dict(bsp=False), False, args),
                                     "domain_bsp": run_inproc(sc4, own4, N, NORMALS_SMOOTH, dict(bsp=True), False, args)}
        ww, wh = weak_film(N, args.width, args.height)
        scw = scenes.soup_weak_scene(args.weak_tris, N, ww, wh)
        r = run_inproc(scw, list(range(N)), N, NORMALS_FLAT, dict(bsp=False), False, args, roofline=True)
        r.update({"workload": WEAK_NOTE % (args.weak_tris, ww, wh), "scaling": "weak", "tiles": N, "tris_per_tile": args.weak_tris, "film": [ww, wh]})
        out["weak_soup"] = r
    out["value"] = out["variants"]["domain_async"]["value"]
    out["unit"] = "Mrays/s"
    print(json.dumps(out))


def spawn_ranks(n):
    """`python bench.py --gpus N` with no launcher around it (the reference's equivalent is one `ibrun -np 2` line, CMakeLists.txt:650-654):
    this process starts N fresh children of the same command line, one per GPU, with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set as torch.distributed.run would, hands rank 0's standard output (the ONE JSON line) through and leaves with the first
    non-zero status of a child.  It imports neither torch nor the library and makes no GPU call; nothing is re-executed: the children are
    ordinary child processes.  When one rank dies the others are given a few seconds (they usually run into the exchange's error path by
    themselves) and are then ended by their exact PIDs."""
    import signal
    import socket

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   GVT_BENCH_SELF_SPAWNED="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))  # rank 0 prints the line; stderr of every rank is inherited
    rc, first_exit, grace = 0, None, float(os.environ.get("GVT_BENCH_SPAWN_GRACE_S", "30"))
    try:
        while any(p.poll() is None for p in procs):
            for p in procs:
                c = p.poll()
                if c is not None and c != 0 and rc == 0:
                    rc, first_exit = c, time.perf_counter()
            if first_exit is not None and time.perf_counter() - first_exit > grace:
                break
            time.sleep(0.05)
    except KeyboardInterrupt:
        rc = 130
    for p in procs:
        if p.poll() is None:
            p.send_signal(signal.SIGTERM)
    for p in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
        if rc == 0 and p.returncode not in (0, None):
            rc = p.returncode
    if rc != 0:
        print("bench.py: launcher: a rank left with status %d" % rc, file=sys.stderr)
    sys.exit(rc if 0 <= rc < 256 else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--tris", type=int, default=10_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--domains", type=int, default=0, help="N=1 only: cut the soup into this many domains (one rank owns them all)")
    ap.add_argument("--scheduler", choices=["domain", "image"], default="domain",
                    help="N>1: domain (default; north_star: the scene cut into one domain per GPU, rays exchanged over RCCL) or image "
                         "(Tracer<ImageScheduler> under several ranks, ImageTracer.h:111-125: scene replicated, camera rays split; the scaling upper bound)")
    ap.add_argument("--bsp", action="store_true", help="N>1: trace until dry before every exchange (Tracer<DomainScheduler>) instead of asynchronous ticks")
    ap.add_argument("--harness", choices=["native", "python", "checker"], default="native",
                    help="native: gvt_hip_tracer (the product path).  python: the Python scheduler loops over the same C ABI (test harness).  "
                         "checker: the Python loops over the CPU checker backend -- no GPU, rehearses this script's N>1 plumbing under gloo")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (gvt_hip_set_option), for experiments")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="N>1: skip the per-variant image check against rank 0's one-rank render of the same scene")
    ap.add_argument("--no-abi-path", action="store_true")
    ap.add_argument("--no-sustained", action="store_true", help="skip the `sustained` key (>= 2000 frames / 2.5 s of the same frame back to back)")
    ap.add_argument("--cpu-row-stride", type=int, default=1)
    ap.add_argument("--same-gpu", action="store_true", help="rehearsal only: every rank uses device 0 (needs an RCCL that tolerates it)")
    ap.add_argument("--fake-comm", action="store_true", help="rehearsal only (one GPU, --same-gpu): no communicator -- every rank renders every scene alone, all domains local; "
                                                              "exercises this script's N>1 native plumbing (variants, extra legs, reductions) where RCCL cannot run; the numbers mean nothing")
    ap.add_argument("--single-variant", action="store_true", help="N>1: only the variant --scheduler / --bsp select (default: Domain asynchronous = `value`, "
                                                                    "plus Domain BSP and the replicated Image scheduler as extra keys of the same line)")
    ap.add_argument("--inproc-ranks", type=int, default=0, help="N=1 only: run the native multi-rank frame loop with this many in-process ranks on ONE GPU "
                                                                 "(hub transport) and print the per-variant tick / byte / phase table; a diagnostic, not a scaling run")
    ap.add_argument("--exchange-timeout-ms", type=int, default=0, help="N>1: deadline of every blocking point of the ray exchange (default: the library's 20 s)")
    ap.add_argument("--no-extra-legs", action="store_true", help="N>1 / --inproc-ranks: skip the extra keys config4_bunny_grid (BASELINE configs[3]: the 8-bunny grid at "
                                                                  "1900x1080, domain d on rank d mod N) and weak_soup (N tiles of --weak-tris triangles each, film scaled with N)")
    ap.add_argument("--weak-tris", type=int, default=10_000_000, help="weak_soup: triangles per tile (= per rank)")
    ap.add_argument("--config4-width", type=int, default=1900)
    ap.add_argument("--config4-height", type=int, default=1080)
    args = ap.parse_args()
    if args.inproc_ranks > 1:
        if int(os.environ.get("WORLD_SIZE", "1")) != 1 or args.gpus != 1:
            print("bench.py: --inproc-ranks is a one-process, one-GPU mode", file=sys.stderr)
            sys.exit(2)
        return inproc_ranks(args)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)  # plain `python bench.py --gpus N`: this process becomes the launcher (it never touches the GPU)

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = 0 if args.same_gpu else int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d (a launcher set it): the two must agree" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    if os.environ.get("GVT_BENCH_DIE_RANK") == str(rank) and world > 1:  # test hook (tests/test_domain_gloo.py): this rank dies before the rendezvous
        sys.exit(7)
    import torch.distributed as dist

    on_gpu = args.harness != "checker"
    if on_gpu and not torch.cuda.is_available():
        print("bench.py: no GPU visible; the adapter has no CPU path", file=sys.stderr)
        sys.exit(3)
    dev = torch.device("cuda", local_rank) if on_gpu else torch.device("cpu")
    if on_gpu:
        torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # rendezvous only (unique id, barriers, the report's sums): the ray exchange is the library's own RCCL communicator.
        # The python harness moves rays through torch.distributed instead: nccl (= RCCL) on GPUs, gloo on the CPU rehearsal.
        dist.init_process_group("nccl" if args.harness == "python" else "gloo", rank=rank, world_size=world,
                                **({"device_id": dev} if args.harness == "python" else {}))

    from gravit_amd import scenes
    from gravit_amd.layouts import NORMALS_FLAT

    capi = None
    if on_gpu:
        from gravit_amd import capi
        from gravit_amd.scheduler import Comm, DomainTracer, ImageTracer, NativeTracer

        capi.init(local_rank)
        for o in args.opt:
            k, v = o.split("=")
            capi.set_option(k, int(v))
    else:
        from gravit_amd.scheduler import DomainTracer, ImageTracer
        from tests.oracle_backend import OracleBackend

    image_split = world > 1 and args.scheduler == "image" and args.harness == "native"
    n_dom = (world if not image_split else 1) if world > 1 else max(1, args.domains)
    scene = scenes.soup_scene(args.tris, args.width, args.height) if n_dom == 1 else scenes.soup_domains_scene(args.tris, n_dom, args.width, args.height)
    own_map = (lambda n: [0] * n) if args.fake_comm else (lambda n: [i % world for i in range(n)])
    owner = own_map(scene.n_inst)
    comm = None
    if args.harness == "native":
        if world > 1 and not args.fake_comm:
            uid = [Comm.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            comm = Comm.rccl(uid[0], rank, world)  # fails unless ncclCommCount == world
            if args.exchange_timeout_ms > 0:
                comm.set_deadline_ms(args.exchange_timeout_ms)
        tracer = NativeTracer(scene, NORMALS_FLAT, owner, comm, replicate=image_split)
        frame_stats = lambda: tracer.stats  # noqa: E731
    else:
        backend = None if on_gpu else OracleBackend(scene, NORMALS_FLAT, [o == rank for o in owner])
        if world == 1 and n_dom == 1:
            tracer = ImageTracer(scene, NORMALS_FLAT, backend=backend)
        else:
            tracer = DomainTracer(scene, owner, dist, torch, dev, NORMALS_FLAT, backend=backend, overlap=not args.bsp)
        frame_stats = lambda: {}  # noqa: E731
    build_ms = sum(a.info()["build_ms"] for a in tracer.backend.adapter_cache.values()) if on_gpu else 0.0
    # the figure above is the process's FIRST build (code objects load, rocPRIM initialises: ~8 ms of it); the same mesh built again
    build_ms_warm = None
    if on_gpu and world == 1 and n_dom == 1 and args.harness == "native":
        from gravit_amd.adapter import HipMeshAdapter
        again = HipMeshAdapter(scene.meshes[0])
        build_ms_warm = again.info()["build_ms"]
        again.close()

    def device_sync():
        if on_gpu:
            capi.synchronize()
            torch.cuda.synchronize()

    def barrier():
        device_sync()
        if world > 1:
            dist.barrier()
        device_sync()

    def frame():
        if args.harness == "native":
            try:
                tracer(bsp=args.bsp, image=image_split and not args.fake_comm)  # includes IceTComposite::composite (to rank 0); the PPM download is not part of the frame
            except capi.GvtHipError as e:
                if world == 1:
                    raise
                die_on_exchange_error(rank, world, "the primary variant", e, tracer.stats)
        else:
            tracer()
            if world > 1:
                tracer.composite(download=False)

    barrier()
    # the tracer's per-scene choices (the frame's route -- small rounds through k_finish or per-hop chains, hops inside the merged launches never / early / always: up to six routes, three timed frames each; the parking
    # threshold: long_auto) settle in untimed frames BEFORE the W warm-up frames, so that neither the warm-up nor the timed steps contain a probing frame
    settle_frames = 26 if (on_gpu and world == 1 and n_dom > 1 and args.harness == "native") else 0
    for _ in range(settle_frames):
        frame()
    # One GPU: the warm-up frames are bracketed kernel class by kernel class (closest hit, long rays, any hit) to find the DOMINANT class; the timed
    # steps then carry HIP events around that class only -- every event pair is a few microseconds of the stream, inside a frame of under a
    # millisecond -- and the other classes are measured in a separate pass of K frames BEHIND the timed region.  Several ranks: all three
    # classes in the timed steps (the per-rank rooflines use them).
    warm_st = None
    if on_gpu and world == 1 and args.warmup > 0:
        capi.stats_reset()
        capi.profile(2)
    for _ in range(args.warmup):
        frame()
    barrier()
    if on_gpu:
        if world == 1 and args.warmup > 0:
            warm_st = capi.stats()
        capi.stats_reset()
        # HIP events on the launch stream around the traversal kernels (2: closest hit, long rays, any hit; 3 / 4: closest / any hit only)
        capi.profile(2 if warm_st is None else (3 if warm_st["ms_closest"] >= warm_st["ms_any"] else 4))
    else:
        tracer.backend.rays_closest = tracer.backend.rays_any = 0
    per_frame, sums = [], {"rays_closest": 0, "rays_any": 0, "rays_sent": 0, "rounds": 0, "chains": 0, "host_syncs": 0, "bytes_sent": 0}
    # (the native tracer's per-frame counters are kept as the library filled them and summed after the loop: a frame is synchronous, so
    # every microsecond of Python between two frames is a microsecond of idle device inside the timed region)
    raw_stats = []
    keep_raw = args.harness == "native"
    t0 = time.perf_counter()
    for _ in range(args.steps):
        f0 = time.perf_counter()
        frame()
        per_frame.append(time.perf_counter() - f0)  # a frame ends with a host synchronisation (the last round's report / the composite)
        raw_stats.append(tracer.frame_stats if keep_raw else frame_stats())
    barrier()
    t1 = time.perf_counter()
    for fs in raw_stats:
        for k, v in (fs.as_dict() if hasattr(fs, "as_dict") else fs).items():
            sums[k] = sums.get(k, 0) + v
    st = capi.stats() if on_gpu else {}
    if on_gpu:
        capi.profile(False)
    kernel_ms_source = "HIP events over the timed steps"
    if warm_st is not None:  # the classes that were not bracketed in the timed steps: a separate pass of K frames behind the timed region, all classes bracketed
        timed_cls = "ms_closest" if warm_st["ms_closest"] >= warm_st["ms_any"] else "ms_any"
        capi.stats_reset()
        capi.profile(2)
        for _ in range(args.steps):
            frame()
        post = capi.stats()
        capi.profile(False)
        for k in ("ms_closest", "ms_any", "ms_long"):
            if k != timed_cls:
                st[k] = post[k]
        other = "launches_any" if timed_cls == "ms_closest" else "launches_closest"
        st[other] = post[other]
        kernel_ms_source = "%s: HIP events over the timed steps; the other classes: HIP events over a separate pass of %d frames behind the timed region" % (timed_cls, args.steps)
    if args.harness != "native":  # the harness counts through the library's own counters (or the checker's)
        sums["rays_closest"] = st.get("rays_closest", getattr(tracer.backend, "rays_closest", 0))
        sums["rays_any"] = st.get("rays_any", getattr(tracer.backend, "rays_any", 0))
        sums["rays_sent"] = getattr(tracer, "rays_sent", 0) * args.steps

    elapsed = t1 - t0
    tot = torch.tensor([float(sums["rays_closest"]), float(sums["rays_any"]), float(sums["rays_sent"]), elapsed], dtype=torch.float64)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        elapsed = float(mx[3].item())
    rays_closest, rays_any, rays_sent = float(tot[0].item()), float(tot[1].item()), float(tot[2].item())
    rays_total = rays_closest + rays_any

    # N > 1: the scheduler variants in ONE invocation -- Domain asynchronous, Domain BSP and (native harness) the replicated Image
    # scheduler -- as extra keys of the line; `value` stays the primary variant measured above
    def reduce_sum(v):
        t = torch.tensor(v, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(x) for x in t]

    def reduce_max(v):
        t = torch.tensor(v, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return [float(x) for x in t]

    # N > 1: every variant and leg is VERIFIED in the same invocation -- rank 0 renders the variant's scene once more with a one-rank tracer of its own
    # (no communicator) and compares the composited framebuffer bit for bit, and the rays traced (summed over the ranks) with the one-rank counts
    do_parity = world > 1 and not args.no_parity and not args.fake_comm
    parity_bad = []  # names of the variants / legs whose image or ray counts differ: the line is still printed, rank 0 leaves with status 4
    ref_cache = {}
    skip_known_user = int(dict(o.split("=") for o in args.opt).get("skip_known", 0))

    def variant_parity(name, backend_v, sc_v, mode_v, tot_v, skip_known=0):
        """rank 0 only (the other ranks go on to the next barrier and wait there)"""
        if not do_parity or rank != 0:
            return None
        t_p = time.perf_counter()
        key = (id(sc_v), mode_v, int(skip_known))
        if key not in ref_cache:
            ref_cache[key] = one_rank_reference(sc_v, mode_v, args.harness, on_gpu, capi, skip_known, skip_known_user) + (sc_v,)  # (the scene stays alive: its id is the key)
        ref_fb, ref_c, ref_a, _ = ref_cache[key]
        got = backend_v.framebuffer(False)
        par = frame_parity(name, got, ref_fb, (tot_v[0] / args.steps, tot_v[1] / args.steps), (ref_c, ref_a))
        if skip_known:  # the opt-in shortcut against the reference's hop-by-hop rule (not image-identical in general): reported, not required
            strict = ref_cache.get((id(sc_v), mode_v, 0))
            if strict is not None:
                a, b = np.asarray(got, np.float32).reshape(-1, 4), np.asarray(strict[0], np.float32).reshape(-1, 4)
                par["vs_strict_rule"] = {"bit_exact": bool(np.array_equal(a, b)), "pixels_differ": int((a != b).any(axis=1).sum()), "max_abs_diff": float(np.abs(a - b).max())}
        par["seconds"] = time.perf_counter() - t_p
        if not (par["bit_exact"] and par["rays_equal"]):
            parity_bad.append(name)
        return par

    variants = None
    if world > 1 and on_gpu and args.harness == "native":
        capi.set_option("frame_timing", 1)  # the variants and the extra legs carry the per-phase breakdown (five more event calls per exchange); `value` above ran without
    legs_error = None  # the first secondary measurement that failed (LegFailed): what is behind it is skipped, the line is still printed
    if world > 1 and not args.single_variant:
        variants = {}
        primary = "image_replicated" if image_split else ("domain_bsp" if args.bsp else "domain_async")
        todo = [("domain_async", "domain", False), ("domain_bsp", "domain", True)] + ([("image_replicated", "image", False)] if args.harness == "native" else [])
        if args.harness == "native" and on_gpu:  # the opt-in known-miss shortcut of shuffleRays (skip_known = 1; not image-identical in general): fewer ticks
            todo.append(("domain_async_known_miss_shortcut", "domain", False))
        for name, kind, bsp_v in todo:
            if legs_error is not None:
                break
            made = None
            shortcut = name.endswith("known_miss_shortcut")
            if shortcut:
                capi.set_option("skip_known", 1)
            if args.harness == "native":
                if (kind == "image") == image_split:
                    tr_v = tracer  # same scene and tracer as the primary variant
                else:
                    sc_v = scenes.soup_scene(args.tris, args.width, args.height) if kind == "image" else scenes.soup_domains_scene(args.tris, world, args.width, args.height)
                    tr_v = made = NativeTracer(sc_v, NORMALS_FLAT, own_map(sc_v.n_inst), comm, replicate=(kind == "image"))
                run_v = (lambda t=tr_v, b=bsp_v, im=(kind == "image" and not args.fake_comm): t(bsp=b, image=im))
                stats_v = (lambda t=tr_v: t.stats)
            else:  # the Python scheduler loops (harness / CPU rehearsal): a tracer per variant, counters through the backend
                be_v = None if on_gpu else OracleBackend(scene, NORMALS_FLAT, [o == rank for o in owner])
                tr_v = made = DomainTracer(scene, owner, dist, torch, dev, NORMALS_FLAT, backend=be_v, overlap=not bsp_v)
                last = {"c": 0, "a": 0}

                def run_v(t=tr_v):
                    t()
                    t.composite(download=False)

                def stats_v(t=tr_v, last=last):
                    c, a = getattr(t.backend, "rays_closest", 0), getattr(t.backend, "rays_any", 0)
                    d = {"rays_closest": c - last["c"], "rays_any": a - last["a"], "rays_sent": getattr(t, "rays_sent", 0), "rounds": getattr(t, "rounds", 0)}
                    last["c"], last["a"] = c, a
                    return d
            try:
                res_v, _, _, tot_v = measure_variant(run_v, stats_v, args.steps, args.warmup, barrier, reduce_sum, reduce_max, rank, world, name)
            except LegFailed as e:
                legs_error = str(e)
                variants[name] = {"failed": legs_error, "is_value": name == primary}
                continue
            finally:
                if shortcut:
                    capi.set_option("skip_known", int(dict(o.split("=") for o in args.opt).get("skip_known", 0)))
            res_v["is_value"] = name == primary
            variants[name] = res_v
            par = variant_parity(name, tr_v.backend, tr_v.scene if args.harness == "native" else scene, NORMALS_FLAT, tot_v, 1 if shortcut else 0)
            if par is not None:
                res_v["parity"] = par
            if made is not None and hasattr(made, "close"):
                made.close()

    primary_parity = None
    if world > 1 and variants is None:  # --single-variant: the one variant that ran is checked like the others
        primary_parity = variant_parity("primary", tracer.backend, scene, NORMALS_FLAT, (rays_closest, rays_any))

    # N > 1, two more keys of the same line (never `value`): BASELINE configs[3] -- the 8-bunny grid under the Domain scheduler, the
    # configuration BASELINE.json names for the scaling curve -- and the weak-scaling soup (N tiles of 10 M triangles each)
    extra = {}
    if world > 1 and not args.no_extra_legs and not args.single_variant and legs_error is None:
        from gravit_amd.layouts import NORMALS_SMOOTH

        def extra_legs():
            def tracer_for(sc, mode, own, bsp_v):
                """(run_frame, frame_stats, close) of one Domain-scheduler variant on scene sc"""
                if args.harness == "native":
                    t = NativeTracer(sc, mode, own, comm)
                    return (lambda composite=True: t(bsp=bsp_v, composite=composite)), (lambda: t.stats), t.close, t.backend
                be = None if on_gpu else OracleBackend(sc, mode, [o == rank for o in own])
                t = DomainTracer(sc, own, dist, torch, dev, mode, backend=be, overlap=not bsp_v)
                last = {"c": 0, "a": 0}

                def run(composite=True):
                    t()
                    if composite:
                        t.composite(download=False)

                def stats():
                    c, a = getattr(t.backend, "rays_closest", 0), getattr(t.backend, "rays_any", 0)
                    if on_gpu:
                        g = capi.stats()
                        c, a = g["rays_closest"], g["rays_any"]
                    d = {"rays_closest": c - last["c"], "rays_any": a - last["a"], "rays_sent": getattr(t, "rays_sent", 0), "rounds": getattr(t, "rounds", 0)}
                    last["c"], last["a"] = c, a
                    return d
                return run, stats, (lambda: None), t.backend

            sc4 = scenes.bunny_grid_scene(width=args.config4_width, height=args.config4_height)
            own4 = own_map(sc4.n_inst)
            leg = {"workload": CONFIG4_NOTE, "film": [args.config4_width, args.config4_height]}
            for name, bsp_v in (("domain_async", False), ("domain_bsp", True)):
                run_v, stats_v, close_v, be_v = tracer_for(sc4, NORMALS_SMOOTH, own4, bsp_v)
                leg[name], _, _, tot4 = measure_variant(run_v, stats_v, args.steps, args.warmup, barrier, reduce_sum, reduce_max, rank, world, "config4_bunny_grid " + name)
                leg[name]["rays_per_step"] = (tot4[0] + tot4[1]) / args.steps
                par = variant_parity("config4_bunny_grid " + name, be_v, sc4, NORMALS_SMOOTH, tot4)
                if par is not None:
                    leg[name]["parity"] = par
                close_v()
            extra["config4_bunny_grid"] = leg
            sc4 = None
            # weak_soup: every rank generates its own tile only; the tiles' boxes are exchanged through the rendezvous group (a Domain-scheduler
            # rank knows every instance's box, DomainTracer.h:115-144)
            ww, wh = weak_film(world, args.width, args.height)
            mine = scenes.soup_weak_scene(args.weak_tris, world, ww, wh, own=None if args.fake_comm else [rank], boxes=[(np.zeros(3), np.zeros(3))] * world)
            boxes = [None] * world
            dist.all_gather_object(boxes, (mine.inst_lo[rank].tolist(), mine.inst_hi[rank].tolist()))
            mine.inst_lo[:] = np.array([b[0] for b in boxes], np.float32)
            mine.inst_hi[:] = np.array([b[1] for b in boxes], np.float32)
            run_v, stats_v, close_v, be_w = tracer_for(mine, NORMALS_FLAT, own_map(world) if args.fake_comm else list(range(world)), False)
            if on_gpu:
                capi.stats_reset()

            barrier()
            for _ in range(args.warmup):
                try:
                    run_v()
                except Exception as e:  # noqa: BLE001
                    if type(e).__name__ != "GvtHipError":
                        raise
                    raise LegFailed("weak_soup (warm-up): %s" % e) from e
            if on_gpu:
                capi.stats_reset(); capi.profile(2)
            wsum = {}

            def stats_w():
                d = stats_v()
                for k in ("rays_closest", "rays_any"):
                    wsum[k] = wsum.get(k, 0) + d.get(k, 0)
                return d

            weak, _, _, totw = measure_variant(run_v, stats_w, args.steps, 0, barrier, reduce_sum, reduce_max, rank, world, "weak_soup")
            roof = None
            if on_gpu:
                stw = capi.stats(); capi.profile(False)
                roof = rank_roofline(stw, wsum.get("rays_closest", 0), wsum.get("rays_any", 0), args.weak_tris)
            roofs = [None] * world
            dist.all_gather_object(roofs, roof)
            weak.update({"workload": WEAK_NOTE % (args.weak_tris, ww, wh), "scaling": "weak", "tiles": world, "tris_per_tile": args.weak_tris, "film": [ww, wh],
                         "rays_per_step": (totw[0] + totw[1]) / args.steps, "roofline_per_rank": roofs})
            if do_parity:
                # no rank holds the whole N x 10 M-triangle scene, so the check is the composite's bookkeeping: every deposit adds 1.0 to a pixel's alpha
                # (IceTComposite::localAdd) -- one more frame WITHOUT the composite gives every rank's own deposits, their sum over the ranks must be what
                # rank 0's composited framebuffer holds, and no rank may have lost a shadow ray (deposits <= shadow rays traced)
                def alpha_sum():
                    device_sync()
                    if on_gpu:
                        a = be_w.fb_tensor(torch, dev).view(-1, 4)[:, 3]
                        r = [float(a.double().sum().item()), float((a > 0).sum().item())]
                        torch.cuda.synchronize()
                        return r
                    a = be_w.framebuffer(False)[..., 3]
                    return [float(a.astype(np.float64).sum()), float((a > 0).sum())]

                try:
                    barrier()
                    run_v(composite=False)
                    stats_v()
                    own = alpha_sum()
                    own_tot = reduce_sum([own[0]])[0]
                    barrier()
                    run_v()
                    d_last = stats_v()
                    any_tot = reduce_sum([float(d_last.get("rays_any", 0))])[0]
                    comp = alpha_sum()
                except Exception as e:  # noqa: BLE001
                    if type(e).__name__ != "GvtHipError":
                        raise
                    raise LegFailed("weak_soup (deposit check): %s" % e) from e
                if rank == 0:
                    ok = comp[0] == own_tot and 0 < own_tot <= any_tot
                    weak["parity"] = {"checked": "deposit counts: alpha of rank 0's composited framebuffer vs the sum over the ranks of their own deposits (a frame without the composite); "
                                                 "no rank holds the whole scene, so there is no one-rank image to compare with",
                                      "deposits_composited": comp[0], "deposits_summed_over_ranks": own_tot, "lit_pixels_composited": int(comp[1]),
                                      "shadow_rays_traced": any_tot, "deposits_equal": bool(ok)}
                    if not ok:
                        parity_bad.append("weak_soup")
            extra["weak_soup"] = weak
            close_v()
            mine = None

        try:
            extra_legs()
        except LegFailed as e:
            legs_error = str(e)

    if rank == 0:
        gpu_fb = tracer.backend.framebuffer(False) if (on_gpu and world == 1 and n_dom == 1) else None  # before the extra legs reuse the backend
        n_tris_local = max(m.tris.shape[0] for m in scene.meshes)
        b_ray = algorithmic_bytes_per_ray(n_tris_local)
        out = {
            "metric": "Mrays/s (primary+shadow) at 1080p, 10M-tri scene",
            "value": rays_total / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_median": float(np.median(per_frame)) * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "soup-%d: %d random triangles (Philox seed 12345, half-extent 0.005), %dx%d, eye (.5,.5,3)->(.5,.5,.5), "
                            "fov 30deg, 1 point light at the eye, depth 1 (primary + 1 shadow ray), AO off"
                            % (args.tris, args.tris, args.width, args.height),
                "scheduler": ("image, 1 domain" if n_dom == 1 else "domain, %d spatial domains on 1 GPU" % n_dom) if world == 1
                             else ("image, scene replicated on %d GPUs, camera rays split (ImageTracer.h:111-125), composite over RCCL" % world) if image_split
                             else "domain, %d spatial domains, 1 per GPU, %s, ray exchange = RCCL p2p issued by libgvt_hip.so"
                                  % (world, "BSP rounds" if args.bsp else "asynchronous ticks"),
                "harness": args.harness,
                "rays_per_step": rays_total / args.steps,
                "primary_traced_per_step": rays_closest / args.steps,
                "shadow_traced_per_step": rays_any / args.steps,
                "rays_sent_per_step": rays_sent / args.steps,
                "rounds_per_step": sums["rounds"] / args.steps, "launch_chains_per_step": sums["chains"] / args.steps,
                "host_syncs_per_step": sums["host_syncs"] / args.steps, "settle_frames": settle_frames,
                "bvh_build_ms": build_ms, "bvh_build_Mtris_per_s": (sum(m.tris.shape[0] for m in scene.meshes) / world / (build_ms * 1e-3) / 1e6) if build_ms else None,
                "bvh_build_ms_warm": build_ms_warm, "bvh_build_Mtris_per_s_warm": (scene.meshes[0].tris.shape[0] / (build_ms_warm * 1e-3) / 1e6) if build_ms_warm else None,
                "normal_mode": "flat",
            },
        }
        if world > 1 and args.harness == "native":
            out["config"]["rccl_comm_count"] = comm.count if comm is not None else None  # == WORLD_SIZE (gvt_hip_comm_create refuses anything else)
            # what moved the rays: RCCL as the process resolves it, a library named by GVT_HIP_RCCL_LIB (tests: a stand-in), or nothing (--fake-comm)
            out["config"]["transport"] = "none (--fake-comm)" if args.fake_comm else ("GVT_HIP_RCCL_LIB=" + os.path.basename(os.environ["GVT_HIP_RCCL_LIB"])
                                                                                       if os.environ.get("GVT_HIP_RCCL_LIB") else "librccl.so.1")
            if args.same_gpu or args.fake_comm or os.environ.get("GVT_HIP_RCCL_LIB"):
                out["config"]["rehearsal"] = "not a measurement: " + ", ".join(x for x in ("--same-gpu" if args.same_gpu else "", "--fake-comm" if args.fake_comm else "",
                                                                                             "GVT_HIP_RCCL_LIB set" if os.environ.get("GVT_HIP_RCCL_LIB") else "") if x)
        if variants is not None:
            out["config"]["bytes_sent_per_step"] = ([v for v in variants.values() if v["is_value"]] or [{}])[0].get("bytes_sent_per_step")
        if variants is not None:
            out["variants"] = variants  # Domain asynchronous / Domain BSP / replicated Image, each: ticks, ms per tick, rays and bytes sent, per-phase ms
        out.update(extra)  # config4_bunny_grid, weak_soup
        if world > 1:
            checked = {}
            if primary_parity is not None:
                checked["primary"] = primary_parity
            for k, v in (variants or {}).items():
                if "parity" in v:
                    checked[k] = v["parity"]
            for k, v in extra.get("config4_bunny_grid", {}).items():
                if isinstance(v, dict) and "parity" in v:
                    checked["config4_bunny_grid " + k] = v["parity"]
            if "parity" in extra.get("weak_soup", {}):
                checked["weak_soup"] = extra["weak_soup"]["parity"]
            out["parity"] = {"checked": "every variant's composited framebuffer on rank 0 against the same scene rendered by ONE rank on rank 0's device, bit for bit, and the rays "
                                        "traced summed over the ranks against the one-rank counts; weak_soup: deposit counts (details: each variant's `parity`)",
                             "bit_exact": (not parity_bad and bool(checked)) if do_parity else None, "failed": parity_bad,
                             "variants": {k: {kk: v.get(kk) for kk in ("bit_exact", "rays_equal", "deposits_equal", "max_abs_diff", "lit_pixels") if kk in v} for k, v in checked.items()},
                             "skipped": None if do_parity else ("--no-parity" if args.no_parity else "--fake-comm")}
        if legs_error is not None:
            out["legs_error"] = legs_error  # a secondary measurement failed in the ray exchange; it and what was behind it are missing, `value` was measured before
        if on_gpu:
            dom = "closest" if st["ms_closest"] >= st["ms_any"] else "any"
            if warm_st is not None:  # the class that was bracketed in the timed steps
                dom = "closest" if warm_st["ms_closest"] >= warm_st["ms_any"] else "any"
            ms_dom = st["ms_%s" % dom]
            n_launch = max(1, st["launches_%s" % dom])
            rays_dom = {"closest": sums["rays_closest"], "any": sums["rays_any"]}[dom]  # this rank's
            achieved = (rays_dom * b_ray / n_launch) / (ms_dom / n_launch / 1e3) / 1e9 if ms_dom > 0 else 0.0
            merged = args.harness == "native" and n_dom > 1  # a one-queue round runs the single-mesh kernels
            symbol = ("k_trace<%s, true, %d, false, true, %s>" % ("false" if dom == "closest" else "true", 0 if dom == "closest" else 1, "true" if merged else "false"))
            traffic, traffic_src = None, None
            tf = os.path.join(ROOT, "profiles", "traffic.json")  # PMC-derived HBM bytes per launch, committed with its provenance
            if os.path.exists(tf):
                try:
                    tj = json.load(open(tf))
                    traffic = tj.get("k_%s_bytes_per_launch" % dom) if (world == 1 and n_dom == 1) else None  # counted on the one-GPU launch only
                    traffic_src = {"file": "profiles/traffic.json", "tag": tj.get("tag"), "profiled_commit": tj.get("commit"),
                                   "profiled_source_hash": tj.get("source_hash"), "kernel": tj.get("k_%s_kernel" % dom)}
                except Exception:
                    traffic = None
            peak_meas = measured_stream_peak(torch, dev)
            out["roofline"] = {
                "bound": "hbm", "kernel": symbol, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                "peak_measured_copy": peak_meas, "frac_of_measured_copy": achieved / peak_meas if peak_meas else None,
                "peak_measured_copy_note": "a 1 GiB torch copy_ (device to device) timed in this run: a floor for the achievable rate, below the guide's 6.29 TB/s float4 copy",
                "algorithmic_bytes_per_ray": b_ray, "rays_per_launch": rays_dom / n_launch, "avg_launch_ms": ms_dom / n_launch,
                "kernel_ms": {k: st[k] for k in ("ms_closest", "ms_any", "ms_shade", "ms_shuffle", "ms_camera", "ms_convert", "ms_sort", "ms_long")},
                "kernel_ms_source": kernel_ms_source,
                "commit": git_head(),  # None on the GPU box (no .git travels); source_hash ties the line to a tree either way
                "source_hash": source_hash(),
            }
            if world == 1 and n_dom == 1:
                try:  # mean visits per primary ray (diagnostic kernel over the binary LBVH; the traversal itself walks its 4-wide collapse)
                    B = tracer.backend
                    B.begin_frame()
                    B.generate_and_filter(None)
                    rr = B.queues[0].to_numpy()[::17]
                    B.begin_frame()
                    vs = B.adapter(0).visit_stats(rr["origin"], rr["direction"])
                    out["roofline"]["visits_per_primary_ray"] = {"inner_nodes_binary": vs["inner_per_ray"], "leaves": vs["leaf_per_ray"],
                                                                  "triangle_tests": vs["tri_tests_per_ray"], "sample_rays": int(len(rr))}
                except Exception as e:
                    out["roofline"]["visits_per_primary_ray"] = "failed: %r" % (e,)
        if on_gpu and world == 1 and n_dom == 1 and args.harness == "native" and not args.no_sustained:
            try:
                out["sustained"] = sustained(frame, rays_total / args.steps)
            except Exception as e:
                out["sustained"] = {"failed": repr(e)}
        if on_gpu and world == 1 and n_dom == 1 and args.harness == "native" and not args.no_abi_path:
            try:
                out["two_frames_in_flight"] = two_frames_in_flight(scene, args.steps, rays_total / args.steps)
            except Exception as e:
                out["two_frames_in_flight"] = {"failed": repr(e)}
        if on_gpu and world == 1 and n_dom == 1 and not args.no_abi_path:
            try:
                out["abi_path"] = abi_path(scene, tracer, capi, np)
            except Exception as e:
                out["abi_path"] = {"failed": repr(e)}
        if on_gpu and world == 1 and n_dom == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(scene, args.cpu_row_stride, host_cores(), gpu_fb=gpu_fb, parity_out=out,
                                                   adapter=tracer.backend.adapter(0) if args.harness == "native" else None)
            except Exception as e:  # the checker is optional for the measurement itself
                out["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        if world > 1 and not args.no_cpu_baseline:
            # the CPU column of the N > 1 line: the same one-GPU workload (the un-cut soup, the whole film) on rank 0's host cores while the other ranks wait at
            # the last barrier; rank 0's own one-rank render of that scene (the image variant's reference) is compared with the CPU oracle's frame on the way
            try:
                one = [v for v in ref_cache.values() if v[3].n_inst == 1 and len(v[3].meshes[0].tris) == args.tris]
                sc1 = one[0][3] if one else scenes.soup_scene(args.tris, args.width, args.height)
                ad1 = None
                if on_gpu and args.harness == "native":
                    from gravit_amd.adapter import HipMeshAdapter
                    ad1 = HipMeshAdapter(sc1.meshes[0])
                side = {}
                out["cpu_baseline"] = cpu_baseline(sc1, args.cpu_row_stride, host_cores(), gpu_fb=one[0][0] if (one and on_gpu) else None, parity_out=side, adapter=ad1)
                out["cpu_baseline"]["note"] = "rank 0's host cores, the one-GPU workload (soup-%d un-cut, %dx%d), after the timed regions of every rank" % (args.tris, args.width, args.height)
                if "parity" in side:
                    out["parity"]["one_rank_render_vs_cpu_oracle"] = side["parity"]
                if ad1 is not None:
                    ad1.close()
            except Exception as e:  # noqa: BLE001
                out["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out), flush=True)
    bad_status = 4 if (rank == 0 and parity_bad) else 0  # an image or a ray count that differs from the one-rank render: the line is out, the run is NOT a success
    if bad_status:
        print("bench.py: parity FAILED for %s (see `parity` in the line)" % ", ".join(parity_bad), file=sys.stderr, flush=True)
    if world > 1:
        if legs_error is not None:  # the communicator was aborted: no tear-down through it, no barrier a missing peer would hang
            sys.stderr.flush()
            os._exit(bad_status)  # (every other rank 0: a non-zero worker would make the launcher stop rank 0 before its line is out)
        dist.barrier()
        dist.destroy_process_group()
    if bad_status:
        sys.exit(bad_status)


if __name__ == "__main__":
    main()
