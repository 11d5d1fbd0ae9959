#!/usr/bin/env python3
"""bench.py -- Mrays/s (primary + shadow) of the GraviT adapter hot path on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): 10,000,000 random triangles,
1920x1080, one point light at the eye, depth 1 (primary + 1 shadow ray), AO off.  A step = one frame:
camera rays -> top-level test -> Adapter::trace (closest hit, shade, shadow rays, any hit) -> shuffle ->
framebuffer, all resident in HBM.  N = 1: Image scheduler, one domain.  N > 1: the same soup cut into N
spatial domains, one per GPU, Domain scheduler with the ray exchange over RCCL (strong scaling: the frame
is fixed).  value = (rays through the closest-hit kernel + rays through the any-hit kernel, all ranks) /
wall time of the timed steps.

  python bench.py --gpus 1 --steps 10 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes_per_ray(n_tris):
    """32 (o,d,tmin,tmax) + 16 (t,prim,u,v) + 32*ceil(log2(T/4)) (one box per level) + 4*48 (one leaf)."""
    levels = max(1, math.ceil(math.log2(max(n_tris, 8) / 4.0)))
    return 32 + 16 + 32 * levels + 4 * 48


def cpu_baseline(scene, row_stride, nthreads, repeats=3, gpu_fb=None, parity_out=None):
    """The CPU oracle (a port of the reference's Embree adapter path; Embree 2.x itself is not in the tree) timed on a
    bounded sample of the same frame: every row_stride-th scanline of the 1080p camera (default: the whole frame), on the
    host cores, best of `repeats` passes."""
    import numpy as np

    from oracle import orc

    cam = scene.camera
    m = scene.meshes[0]
    om = orc.Mesh(m.verts, m.tris, mesh_mat=m.material)  # BVH build excluded, like on the GPU side
    rays = orc.camera_rays(cam.eye, cam.focus, cam.up, cam.fov, cam.width, cam.height)
    rows = np.arange(0, cam.height, row_stride)
    sel = (rows[:, None] * cam.width + np.arange(cam.width)[None, :]).reshape(-1)
    sample = rays[sel].copy()
    nxt, tt = orc.toplevel_intersect(scene.inst_lo, scene.inst_hi, [0], sample)
    hit = nxt >= 0
    s2 = sample[hit].copy()
    s2["origin"] += s2["direction"] * (tt[hit] * np.float32(0.95))[:, None]
    dt = None
    moved = None
    for _ in range(repeats):
        rays_in = s2.copy()
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
    return {"value": (c + a) / dt / 1e6, "unit": "Mrays/s", "cores": nthreads, "kind": "port",
            "sample": "scanlines 0,%d,.. of the %dx%d frame: %d primary + %d shadow rays in %.3f s wall on %d threads (best of %d; "
                      "%.0f core-seconds per pass), BVH build excluded; CPU oracle = port of the Embree adapter path (Embree 2.x not in the tree)"
                      % (row_stride, cam.width, cam.height, c, a, dt, nthreads, repeats, dt * nthreads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--tris", type=int, default=10_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--domains", type=int, default=0, help="N=1 only: cut the soup into this many domains and use the Domain scheduler")
    ap.add_argument("--full-reduce", action="store_true", help="N>1: composite by a sum-reduce of whole frames instead of each rank's written rectangle")
    ap.add_argument("--bsp", action="store_true", help="N>1: bulk-synchronous exchange rounds instead of the overlapped exchange")
    ap.add_argument("--opt", action="append", default=[], help="library option name=value (gvt_hip_set_option), for experiments")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-row-stride", type=int, default=1)
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    import torch.distributed as dist

    if not torch.cuda.is_available():
        print("bench.py: no GPU visible; the adapter has no CPU path", file=sys.stderr)
        sys.exit(3)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from gravit_amd import capi, scenes
    from gravit_amd.layouts import NORMALS_FLAT
    from gravit_amd.scheduler import DomainTracer, ImageTracer

    capi.init(local_rank)
    for o in args.opt:
        k, v = o.split("=")
        capi.set_option(k, int(v))
    if world == 1:
        capi.set_stream(torch.cuda.current_stream().cuda_stream)
    # world > 1: the adapter keeps its own non-blocking stream, so that RCCL transfers posted on torch's side really run beside the
    # traversal kernels; the scheduler synchronises explicitly where the two meet (DomainTracer._post_exchange / _complete_exchange)

    if world == 1 and args.domains > 1:
        scene = scenes.soup_domains_scene(args.tris, args.domains, args.width, args.height)
        tracer = DomainTracer(scene, [0] * scene.n_inst, dist, torch, dev, NORMALS_FLAT)
    elif world == 1:
        scene = scenes.soup_scene(args.tris, args.width, args.height)
        tracer = ImageTracer(scene, NORMALS_FLAT)
    else:
        scene = scenes.soup_domains_scene(args.tris, world, args.width, args.height)
        owner = [i % world for i in range(scene.n_inst)]
        tracer = DomainTracer(scene, owner, dist, torch, dev, NORMALS_FLAT, overlap=not args.bsp)
    build_ms = sum(a.info()["build_ms"] for a in tracer.backend.adapter_cache.values())

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def frame():
        tracer()
        if world > 1:
            tracer.composite(download=False, rows_only=not args.full_reduce)  # IceTComposite::composite is part of the frame, the PPM download is not

    for _ in range(args.warmup):
        frame()
    barrier()
    capi.stats_reset()
    capi.profile(True)  # HIP events around every kernel, on the launch stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        frame()
    capi.synchronize()
    barrier()
    t1 = time.perf_counter()
    st = capi.stats()
    capi.profile(False)

    elapsed = t1 - t0
    tot = torch.tensor([float(st["rays_closest"]), float(st["rays_any"]), elapsed], dtype=torch.float64, device=dev)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        elapsed = float(mx[2].item())
    rays_closest, rays_any = float(tot[0].item()), float(tot[1].item())
    rays_total = rays_closest + rays_any

    if rank == 0:
        n_tris_local = max(m.tris.shape[0] for m in scene.meshes)
        b_ray = algorithmic_bytes_per_ray(n_tris_local)
        dom = "closest" if st["ms_closest"] >= st["ms_any"] else "any"
        ms_dom = st["ms_%s" % dom]
        n_launch = max(1, st["launches_%s" % dom])
        rays_dom = st["rays_%s" % dom]
        achieved = (rays_dom * b_ray / n_launch) / (ms_dom / n_launch / 1e3) / 1e9 if ms_dom > 0 else 0.0
        traffic = None
        tf = os.path.join(ROOT, "profiles", "traffic.json")  # PMC-derived HBM bytes per launch, committed with its provenance
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("k_%s_bytes_per_launch" % dom)
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s (primary+shadow) at 1080p, 10M-tri scene",
            "value": rays_total / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": "soup-%d: %d random triangles (Philox seed 12345, half-extent 0.005), %dx%d, eye (.5,.5,3)->(.5,.5,.5), "
                            "fov 30deg, 1 point light at the eye, depth 1 (primary + 1 shadow ray), AO off"
                            % (args.tris, args.tris, args.width, args.height),
                "scheduler": ("image (1 domain)" if args.domains <= 1 else "domain (%d spatial domains on 1 GPU)" % args.domains) if world == 1 else "domain (%d spatial domains, 1 per GPU, RCCL p2p ray exchange)" % world,
                "rays_per_step": rays_total / args.steps,
                "primary_traced_per_step": rays_closest / args.steps,
                "shadow_traced_per_step": rays_any / args.steps,
                "bvh_build_ms": build_ms,
                "normal_mode": "flat",
            },
            "roofline": {
                "bound": "hbm", "kernel": "k_%s" % dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_ray": b_ray, "rays_per_launch": rays_dom / n_launch, "avg_launch_ms": ms_dom / n_launch,
                "kernel_ms": {k: st[k] for k in ("ms_closest", "ms_any", "ms_shade", "ms_shuffle", "ms_camera", "ms_convert", "ms_sort", "ms_long")},
            },
        }
        if world == 1 and args.domains <= 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(scene, args.cpu_row_stride, os.cpu_count() or 1, gpu_fb=tracer.backend.framebuffer(False), parity_out=out)
            except Exception as e:  # the checker is optional for the measurement itself
                out["cpu_baseline"] = {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (e,)}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
