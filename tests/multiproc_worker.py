"""One RANK of tests/test_gpu_multiproc.py: a process of its own on the box's one GPU, its communicator built by the library's RCCL leg
(gvt_hip_comm_create -> ncclCommInitRank, grouped ncclSend / ncclRecv per tick, ncclReduce for the composite) over the stand-in
tests/fake_rccl (GVT_HIP_RCCL_LIB) -- RCCL itself refuses two ranks on one device.  Rank 0 compares every composited image with the
checker's restated DomainTracer and writes the verdict; a rank that fails exits non-zero.
   usage: RANK=r WORLD_SIZE=n MASTER_ADDR=127.0.0.1 MASTER_PORT=p python tests/multiproc_worker.py <out.json>"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import numpy as np
    import torch.distributed as dist

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)  # rendezvous only: the unique id, the gathers of this script
    from gravit_amd import capi, scenes
    from gravit_amd.layouts import NORMALS_FLAT, NORMALS_SMOOTH
    from gravit_amd.scheduler import Comm, NativeTracer
    from tests import helpers
    from tests.helpers import oracle_render_domain

    helpers.DEFAULT_RULE = "strict"  # like tests/conftest.py for the GPU tests: the reference's hop-by-hop rule, checked strictly
    capi.init(0)
    uid = [Comm.unique_id() if rank == 0 else None]
    dist.broadcast_object_list(uid, src=0)
    comm = Comm.rccl(uid[0], rank, world)
    assert comm.rank == rank and comm.world == world and comm.count == world

    def config5(size, n_dom):
        one = scenes.cathedral_scene(size, size, samples=2, depth=2, eye=(0.0, 1.5, 13.0), light=(0.0, 2.5, 12.0))
        return scenes.split_into_domains(one, n_dom)

    cases = [  # (name, scene, normals, BSP rounds?, options, tolerance of the colour sums)
        ("config4 asynchronous", scenes.bunny_grid_scene(width=380, height=216), NORMALS_SMOOTH, False, {}, 0.0),
        ("config4 BSP", scenes.bunny_grid_scene(width=380, height=216), NORMALS_SMOOTH, True, {}, 0.0),
        ("config5 in 4 slabs, payloads on the communicator's stream", config5(256, 4), NORMALS_FLAT, False, {"payload_overlap_kb": 0}, 1e-5),
        ("config5 in 4 slabs, nothing inline", config5(256, 4), NORMALS_FLAT, True, {"inline_kb": 0}, 1e-5),
        ("soup in 4 tiles", scenes.soup_domains_scene(200_000, 4, 320, 180), NORMALS_FLAT, False, {}, 0.0),
    ]
    report = []
    for name, sc, mode, bsp, opts, tol in cases:
        for k, v in opts.items():
            capi.set_option(k, v)
        owner = [i % world for i in range(sc.n_inst)]
        tr = NativeTracer(sc, mode, owner, comm)
        fb = tr(bsp=bsp).framebuffer(True) if rank == 0 else (tr(bsp=bsp), None)[1]
        st1 = dict(tr.stats)
        fb2 = tr(bsp=bsp).framebuffer(True) if rank == 0 else (tr(bsp=bsp), None)[1]  # the same tracer again: queues, tables and sequence numbers carry over
        stats = [None] * world
        dist.all_gather_object(stats, st1)
        tr.close()
        for k in opts:
            capi.set_option(k, {"payload_overlap_kb": 1024, "inline_kb": 16}[k])
        if rank == 0:
            ref, st = oracle_render_domain(sc, owner, world, mode)
            assert (ref[..., :3].sum(axis=2) > 0).sum() > 500, name
            err = float(np.abs(fb[..., :3] - ref[..., :3]).max())
            assert err <= tol, (name, err)
            assert np.array_equal(fb[..., 3], ref[..., 3]), name  # deposits per pixel
            assert float(np.abs(fb2 - fb).max()) <= tol, name
            sent = sum(s["rays_sent"] for s in stats)
            assert sent == st.rays_sent and sent > 0, (name, sent, st.rays_sent)
            assert sum(s["rays_closest"] for s in stats) == st.rays_closest and sum(s["rays_any"] for s in stats) == st.rays_any, name
            report.append({"case": name, "max_abs_err": err, "rays_sent": sent, "ticks": [s["rounds"] for s in stats], "exchanges": [s.get("exchanges", 0) for s in stats],
                           "rays_inline": sum(s.get("rays_inline", 0) for s in stats)})
        dist.barrier()
    # the replicated Image scheduler on several ranks: no ray changes rank, the frame ends with the composite (ncclReduce)
    sc = scenes.bunny_grid_scene(width=380, height=216)
    tr = NativeTracer(sc, NORMALS_SMOOTH, [0] * sc.n_inst, comm, replicate=True)
    B = tr(image=True)
    if rank == 0:
        from tests.helpers import oracle_render
        ref, _ = oracle_render(sc, NORMALS_SMOOTH)
        fb = B.framebuffer(True)
        err = float(np.abs(fb[..., :3] - ref[..., :3]).max())
        assert err <= 1e-5 and np.array_equal(fb[..., 3], ref[..., 3]), ("image scheduler", err)
        report.append({"case": "image scheduler, replicated", "max_abs_err": err})
    tr.close()
    dist.barrier()
    # the reference's own distributed CTest (`ibrun -np 2 gvtSimple / gvtFileLoad -image / -domain`, CMakeLists.txt:644-688): rank 0's PPM against the
    # golden image, gvtImageDiff -tolerance 300
    from tests.conftest import GOLDEN, read_ppm
    for gname, builder in (("simple", scenes.simple_scene), ("bunny", scenes.bunny_scene)):
        for sched in ("image", "domain"):
            sc = builder()
            image = sched == "image"
            tr = NativeTracer(sc, NORMALS_SMOOTH, [0] * sc.n_inst if image else [i % world for i in range(sc.n_inst)], comm, replicate=image)
            B = tr(bsp=not image, image=image)
            if rank == 0:
                gold = read_ppm(os.path.join(GOLDEN, "ref_%s.ppm" % gname)).astype(np.int64)
                diff = int(np.abs(B.fb.ppm_bytes().astype(np.int64).reshape(gold.shape) - gold).sum())
                assert diff < 300, (gname, sched, diff)
                report.append({"case": "reference CTest: %s -%s, %d ranks" % (gname, sched, world), "ppm_sum_abs_diff": diff})
            tr.close()
            dist.barrier()
    comm.close()
    if rank == 0:
        with open(sys.argv[1], "w") as f:
            json.dump({"world": world, "cases": report}, f)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
