"""Domain scheduler on the device.  The GPU box has ONE MI355X, so the P ranks of Tracer<DomainScheduler> run as P
threads of one process, each with its own HipBackend (own queues, framebuffer, adapters for the domains it owns), and
exchange real device wire buffers (80-byte rays) through an in-process stand-in for torch.distributed
(tests/fake_dist.py).  Everything else -- DomainTracer, HipBackend, the wire conversion kernels, the framebuffer view
used by the composite reduce -- is the product code that runs under RCCL on 8 GPUs."""
import threading

import numpy as np
import pytest
import torch

from gravit_amd import scenes
from gravit_amd.layouts import NORMALS_FLAT, NORMALS_SMOOTH
from gravit_amd.scheduler import DomainTracer, HipBackend
from tests.fake_dist import FakeWorld
from tests.helpers import oracle_render, oracle_render_domain

pytestmark = pytest.mark.gpu


class LockedBackend:
    """libgvt_hip.so serialises work on one stream and shares scratch arenas: one thread inside it at a time."""
    lock = threading.Lock()

    def __init__(self, inner):
        self._b = inner

    def __getattr__(self, name):
        attr = getattr(self._b, name)
        if not callable(attr):
            return attr

        def call(*a, **k):
            with LockedBackend.lock:
                return attr(*a, **k)
        return call


def run_ranks(scene, owner, world, mode, overlap=False):
    fw = FakeWorld(world)
    out = {}
    errs = []

    def rank_main(rank):
        try:
            dist = fw.rank_view(rank)
            owned = [o == rank for o in owner]
            with LockedBackend.lock:
                backend = HipBackend(scene, mode, owned)
            tr = DomainTracer(scene, owner, dist, torch, torch.device("cuda", 0), mode, backend=LockedBackend(backend), overlap=overlap)
            tr()
            fb = tr.composite()
            out[rank] = (fb, tr.rays_sent, tr.rounds, tr.adapter_calls)
        except Exception as e:  # noqa: BLE001
            import traceback
            errs.append(traceback.format_exc())
            try:
                fw.barrier.abort()
            except Exception:
                pass

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=600) for t in th]
    assert not errs, errs[0]
    return out


@pytest.mark.parametrize("world,overlap", [(2, False), (4, False), (8, False), (2, True), (4, True), (8, True)])
def test_bunny_grid_domains_equal_the_single_rank_image(hip, world, overlap):
    """BASELINE config 4 (reduced film): 8 bunny instances, one domain per virtual rank round-robin; BSP rounds and the overlapped
    exchange (several local domains per rank: transfers are in flight during adapter calls)."""
    sc = scenes.bunny_grid_scene(width=380, height=216)
    owner = [i % world for i in range(sc.n_inst)]
    res = run_ranks(sc, owner, world, NORMALS_SMOOTH, overlap)
    fb = res[0][0]
    ref, st = oracle_render_domain(sc, owner, world, 1)
    assert np.array_equal(fb[..., :3], ref[..., :3])
    assert sum(r[1] for r in res.values()) == st.rays_sent and st.rays_sent > 0
    one, _ = oracle_render(sc, 1)
    assert np.array_equal(fb[..., :3], one[..., :3])  # pass criterion of config 4: equal to the 1-GPU image


@pytest.mark.parametrize("size,n_dom,world,overlap", [(1024, 4, 4, False), (384, 8, 8, True), (384, 4, 2, True), (384, 8, 3, False)])
def test_config5_cathedral_domains_on_virtual_ranks(hip, size, n_dom, world, overlap):
    """BASELINE config 5 stand-in under the Domain scheduler: the hall cut into 4 / 8 slabs along its axis, samples = 2
    (4 rays per pixel), depth 2; bounce rays and shadow rays cross slabs and ranks carrying their RNG stream word on the wire.
    Equal to the oracle's restated DomainTracer (max-abs <= 1e-5: several deposits per pixel meet in float atomics and in the
    composite), with the same number of rays sent and the same deposit count per pixel."""
    one = scenes.cathedral_scene(size, size, samples=2, depth=2, eye=(0.0, 1.5, 13.0), light=(0.0, 2.5, 12.0))
    sc = scenes.split_into_domains(one, n_dom)
    owner = [i % world for i in range(sc.n_inst)]
    res = run_ranks(sc, owner, world, NORMALS_FLAT, overlap)
    fb = res[0][0]
    ref, st = oracle_render_domain(sc, owner, world, 0)
    assert (ref[..., :3].sum(axis=2) > 0).mean() > 0.2
    assert np.abs(fb[..., :3] - ref[..., :3]).max() <= 1e-5
    assert np.array_equal(fb[..., 3], ref[..., 3])
    assert sum(r[1] for r in res.values()) == st.rays_sent and st.rays_sent > 10_000
    assert sum(r[3] for r in res.values()) == st.adapter_calls or overlap  # the overlapped loop interleaves calls differently


def test_soup_spatial_domains_with_cross_traffic(hip):
    sc = scenes.soup_domains_scene(200_000, 4, 320, 180)
    sc.camera.eye, sc.camera.focus = (3.0, 0.6, 0.4), (0.5, 0.5, 0.5)  # along -x: rays cross the x-tiled domains
    sc.lights["position"] = (2.0, 2.5, 1.5)
    owner = [i % 2 for i in range(sc.n_inst)]
    res = run_ranks(sc, owner, 2, NORMALS_FLAT)
    ref, st = oracle_render_domain(sc, owner, 2, 0)
    assert np.array_equal(res[0][0][..., :3], ref[..., :3])
    assert sum(r[1] for r in res.values()) == st.rays_sent and st.rays_sent > 1000


def test_wire_roundtrip_and_framebuffer_view(hip):
    """export_wire / append_wire carry the reference's 80-byte Ray image; fb_tensor aliases the device framebuffer."""
    sc = scenes.bunny_grid_scene(width=190, height=108)
    B = HipBackend(sc, NORMALS_FLAT)
    B.begin_frame()
    B.generate_and_filter(None)
    sizes = B.queue_sizes()
    before = [q.to_numpy() for q in B.queues]
    dev = torch.device("cuda", 0)
    insts = [i for i in range(sc.n_inst) if sizes[i]]
    buf = B.export_wire(insts, torch, dev)
    B.sync()
    assert buf.shape == (sum(sizes), 20) and B.queue_sizes() == [0] * sc.n_inst
    host = buf.cpu().numpy().view(np.uint8).reshape(-1, 80)
    off = 0
    for i in insts:
        exp = before[i].view(np.uint8).reshape(-1, 80)
        assert (host[off:off + sizes[i], :64] == exp[:, :64]).all()
        B.append_wire(i, buf, off, sizes[i])
        off += sizes[i]
    B.sync()
    assert B.queue_sizes() == sizes
    for i in insts:
        assert (B.queues[i].to_numpy().view(np.uint8).reshape(-1, 80)[:, :64] == before[i].view(np.uint8).reshape(-1, 80)[:, :64]).all()
    t = B.fb_tensor(torch, dev)
    assert t.numel() == 190 * 108 * 4 and float(t.abs().sum()) == 0.0
    t[5] = 0.75
    torch.cuda.synchronize()
    assert B.framebuffer(False).reshape(-1)[5] == np.float32(0.75)


def test_single_rank_nccl_process_group(hip, tmp_path):
    """RCCL itself, as far as one GPU allows: init, the count all-gather and the framebuffer reduce of DomainTracer."""
    import torch.distributed as dist

    store = dist.FileStore(str(tmp_path / "store"), 1)
    dist.init_process_group("nccl", store=store, rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        sc = scenes.bunny_grid_scene(width=190, height=108)
        tr = DomainTracer(sc, [0] * sc.n_inst, dist, torch, torch.device("cuda", 0), NORMALS_SMOOTH)
        tr()
        mine = torch.tensor(tr.backend.queue_sizes(), dtype=torch.int64, device="cuda")
        rows = [torch.empty_like(mine)]
        dist.all_gather(rows, mine)
        assert int(rows[0].sum()) == 0
        t = tr.backend.fb_tensor(torch, torch.device("cuda", 0))
        dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
        torch.cuda.synchronize()
        ref, _ = oracle_render(sc, 1)
        assert np.array_equal(tr.backend.framebuffer(True)[..., :3], ref[..., :3])
    finally:
        dist.destroy_process_group()
