"""The schedulers as native loops (gvt_hip_tracer, csrc/domain.hip): rounds of ONE merged launch chain over all local queues, the
Domain scheduler's ray exchange through the library's own transport.  On the one-GPU box the ranks of Tracer<DomainScheduler> are
threads of one process, each with its own context (stream, scratch, counters), joined by the in-process transport -- the same
protocol (announce + payload in the reference's wire format, termination from the announces) that runs over RCCL on 8 GPUs; RCCL
itself is exercised as far as one GPU allows (communicator of one rank)."""
import os
import threading

import numpy as np
import pytest

from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_FLAT, NORMALS_SMOOTH
from gravit_amd.scheduler import Comm, Context, HipBackend, ImageTracer, NativeTracer
from tests.helpers import oracle_render, oracle_render_domain

pytestmark = pytest.mark.gpu


def config5(size, n_dom):
    one = scenes.cathedral_scene(size, size, samples=2, depth=2, eye=(0.0, 1.5, 13.0), light=(0.0, 2.5, 12.0))
    return one if n_dom <= 1 else scenes.split_into_domains(one, n_dom)


@pytest.mark.parametrize("name,mode,exact", [("simple", NORMALS_SMOOTH, True), ("bunny_grid", NORMALS_SMOOTH, True), ("bunny_conf", NORMALS_SMOOTH, True),
                                             ("soup", NORMALS_FLAT, True), ("soup4", NORMALS_FLAT, True), ("config5", NORMALS_FLAT, False),
                                             ("config5x4", NORMALS_FLAT, False), ("grid96", NORMALS_SMOOTH, True)])
def test_rounds_equal_the_reference_order_loop(hip, name, mode, exact):
    """Image scheduler, one rank: the round loop (all non-empty queues per launch chain) gives the image of the reference's
    fullest-queue-first loop as restated by the oracle -- bit for bit where a pixel receives one deposit, 1e-5 where several meet."""
    import os

    from tests.conftest import GOLDEN
    sc = {"simple": lambda: scenes.simple_scene(256, 256), "bunny_grid": lambda: scenes.bunny_grid_scene(width=475, height=270),
          "bunny_conf": lambda: scenes.load_conf(os.path.join(GOLDEN, "bunny.conf"), width=475, height=270),
          "soup": lambda: scenes.soup_scene(200_000, 320, 180), "soup4": lambda: scenes.soup_domains_scene(200_000, 4, 320, 180),
          "config5": lambda: config5(512, 1), "config5x4": lambda: config5(384, 4),
          "grid96": lambda: scenes.bunny_grid_scene(nx=12, ny=8, pitch=0.22, width=240, height=160)}[name]()
    tr = NativeTracer(sc, mode)
    B = tr()
    fb = B.framebuffer(True)
    ref, st = oracle_render(sc, mode, nthreads=8)
    assert (ref[..., :3].sum(axis=2) > 0).sum() > 500
    if exact:
        assert np.array_equal(fb[..., :3], ref[..., :3])
    else:
        assert np.abs(fb[..., :3] - ref[..., :3]).max() <= 1e-5
    assert np.array_equal(fb[..., 3], ref[..., 3])
    assert tr.stats["rays_closest"] == st.rays_closest and tr.stats["rays_any"] == st.rays_any
    assert tr.stats["host_syncs"] <= tr.stats["chains"] + 1  # one per round (+ the camera filter unless its counts stay on the device)
    # a second frame of the same tracer: same image (queues, scratch and counters are reused)
    fb2 = tr().framebuffer(True)
    assert np.abs(fb2 - fb).max() <= (0.0 if exact else 1e-5)
    tr.close()


def run_native_ranks(scene, owner, world, mode, bsp, full_reduce=False, image=False, opts=()):
    hub = capi.load().gvt_hip_hub_create(world)
    out, errs = {}, []

    def rank_main(rank):
        ctx = None
        try:
            ctx = Context(0)
            for k, v in opts:  # (knobs belong to the rank's context)
                capi.set_option(k, v)
            comm = Comm.local(hub, rank)
            tr = NativeTracer(scene, mode, owner, comm, replicate=image)
            B = tr(bsp=bsp, full_reduce=full_reduce, image=image)
            out[rank] = (B.framebuffer(True) if rank == 0 else None, dict(tr.stats, reserved_cus=comm.reserved_cus), B.fb.ppm_bytes().copy() if rank == 0 else None)
            tr.close()
            comm.close()
            B = tr = None
        except Exception:  # noqa: BLE001
            import traceback
            errs.append(traceback.format_exc())
            capi.load().gvt_hip_hub_abort(hub)
        finally:
            import gc
            gc.collect()
            if ctx is not None:
                ctx.close()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=600) for t in th]
    capi.load().gvt_hip_hub_destroy(hub)
    assert not errs, errs[0]
    return out


@pytest.mark.parametrize("world,bsp", [(2, True), (4, True), (8, True), (2, False), (4, False), (8, False), (3, False)])
def test_native_domain_scheduler_config4(hip, world, bsp):
    """BASELINE config 4 (reduced film): 8 bunny instances, one domain per rank round-robin, BSP rounds (Tracer<DomainScheduler>) and
    asynchronous ticks: the composited image equals the oracle's restated DomainTracer and the 1-rank image; rays sent agree."""
    sc = scenes.bunny_grid_scene(width=380, height=216)
    owner = [i % world for i in range(sc.n_inst)]
    res = run_native_ranks(sc, owner, world, NORMALS_SMOOTH, bsp)
    fb = res[0][0]
    ref, st = oracle_render_domain(sc, owner, world, 1)
    assert np.array_equal(fb[..., :3], ref[..., :3])
    one, _ = oracle_render(sc, 1)
    assert np.array_equal(fb[..., :3], one[..., :3])
    assert sum(r[1]["rays_sent"] for r in res.values()) == st.rays_sent and st.rays_sent > 0
    assert sum(r[1]["rays_closest"] for r in res.values()) == st.rays_closest
    for r in res.values():  # ONE host synchronisation per local chain / exchange round (+ camera filter + composite)
        assert r[1]["host_syncs"] <= r[1]["rounds"] + r[1]["chains"] + 2
    if bsp:
        assert all(r[1]["rounds"] == st.rounds for r in res.values())


@pytest.mark.parametrize("size,n_dom,world,bsp,overlap_kb", [(768, 4, 4, True, 1024), (384, 8, 8, False, 1024), (384, 4, 2, False, 0), (256, 8, 3, True, 0)])
def test_native_domain_scheduler_config5(hip, size, n_dom, world, bsp, overlap_kb):
    """BASELINE config 5 stand-in (4 rays per pixel, depth 2) under the native Domain scheduler: bounce and shadow rays cross slabs
    and ranks in the reference's wire format with their RNG stream word.  overlap_kb = 0: every payload moves on the communicator's
    own stream beside the next chain (default: only payloads of a megabyte or more; the others stay on the compute stream)."""
    sc = config5(size, n_dom)
    owner = [i % world for i in range(sc.n_inst)]
    res = run_native_ranks(sc, owner, world, NORMALS_FLAT, bsp, opts=(("payload_overlap_kb", overlap_kb),))
    fb = res[0][0]
    ref, st = oracle_render_domain(sc, owner, world, 0)
    assert (ref[..., :3].sum(axis=2) > 0).mean() > 0.2
    assert np.abs(fb[..., :3] - ref[..., :3]).max() <= 1e-5
    assert np.array_equal(fb[..., 3], ref[..., 3])
    assert sum(r[1]["rays_sent"] for r in res.values()) == st.rays_sent and st.rays_sent > 10_000
    assert sum(r[1]["rays_closest"] for r in res.values()) == st.rays_closest
    assert sum(r[1]["rays_any"] for r in res.values()) == st.rays_any


@pytest.mark.parametrize("name,builder", [("simple", scenes.simple_scene), ("bunny", scenes.bunny_scene)])
@pytest.mark.parametrize("scheduler,world", [("image", 1), ("domain", 1), ("image", 2), ("domain", 2)])
def test_reference_ctest_matrix_on_the_native_schedulers(hip, name, builder, scheduler, world):
    """The reference's own CTest (CMakeLists.txt:644-688): gvtSimple / gvtFileLoad bunny.obj, -image and -domain, alone and as `ibrun -np 2`
    (:650-654), each image against Test/CTESTtest/data/<name>.ppm with gvtImageDiff -tolerance 300 (sum of |byte differences| over the file) -- here
    through the native schedulers on one rank and on two in-process ranks (two PROCESSES over the RCCL leg: tests/test_gpu_multiproc.py).
    Domain: mpiInstanceMap round-robin (DomainTracer.h:130-142); Image: the scene on every rank, the camera's list split (ImageTracer.h:111-125)."""
    from tests.conftest import GOLDEN, read_ppm
    sc = builder()
    image = scheduler == "image"
    gold = read_ppm(os.path.join(GOLDEN, "ref_%s.ppm" % name)).astype(np.int64)
    if world == 1:
        tr = NativeTracer(sc, NORMALS_SMOOTH) if image else NativeTracer(sc, NORMALS_SMOOTH, [0] * sc.n_inst, None)
        ppm = tr(bsp=not image).fb.ppm_bytes().copy()  # IceTComposite::write (:131-157), by the library
        tr.close()
    else:
        owner = [0] * sc.n_inst if image else [i % world for i in range(sc.n_inst)]
        ppm = run_native_ranks(sc, owner, world, NORMALS_SMOOTH, bsp=not image, image=image)[0][2]
    assert np.abs(ppm.astype(np.int64).reshape(gold.shape) - gold).sum() < 300


def _camera_path(sc):
    """a camera that moves between frames: sideways and closer (a larger film rectangle: queues grow), looking away (a frame without a single ray),
    back to the start, another field of view, 2x2 samples (four times the rays), depth 2 (bounces: RNG stream words), and the start again"""
    from dataclasses import replace
    c = sc.camera
    eye, foc = np.array(c.eye, np.float64), np.array(c.focus, np.float64)
    view = foc - eye
    side = np.cross(view, np.array(c.up, np.float64)); side /= np.linalg.norm(side)
    return [c,
            replace(c, eye=tuple(eye + 0.35 * view + 0.2 * np.linalg.norm(view) * side), focus=tuple(foc)),
            replace(c, focus=tuple(eye - view)),
            c,
            replace(c, fov=float(c.fov) * 0.6),
            replace(c, samples=2),
            replace(c, depth=2),
            c]


@pytest.mark.parametrize("name,world", [("grid", 1), ("grid", 2), ("bunny", 1)])
def test_a_moving_camera_reuses_the_tracer(hip, name, world):
    """gvt_hip_tracer_set_camera between frames (an interactive GraviT application moves its camera every frame; the tracer, its queues, tables and
    framebuffer stay): every frame of the path equals the checker's image for that camera -- one rank (rounds; "bunny": the lean one-instance frame)
    and two in-process ranks (Domain scheduler); the frames that return to the first camera equal the first frame bit for bit."""
    from dataclasses import replace
    sc = scenes.bunny_grid_scene(width=380, height=216) if name == "grid" else scenes.bunny_scene(256, 256)
    path = _camera_path(sc)
    owner = [i % world for i in range(sc.n_inst)]
    refs = []
    for cam in path:
        s2 = replace(sc, camera=cam)
        refs.append(oracle_render(s2, NORMALS_SMOOTH)[0] if world == 1 else oracle_render_domain(s2, owner, world, NORMALS_SMOOTH)[0])
    assert (refs[0][..., 3] > 0).sum() > 500 and not refs[2][..., 3].any() and refs[5][..., 3].max() >= 4
    frames, errs = {}, []
    hub = capi.load().gvt_hip_hub_create(world) if world > 1 else None

    def rank_main(rank):
        ctx = None
        try:
            ctx = Context(0) if world > 1 else None
            comm = Comm.local(hub, rank) if world > 1 else None
            tr = NativeTracer(sc, NORMALS_SMOOTH, owner, comm)
            out = []
            for cam in path:
                tr.set_camera(cam)
                B = tr(bsp=False)
                out.append(B.framebuffer(True).copy() if rank == 0 else None)
            frames[rank] = out
            tr.close()
            if comm is not None:
                comm.close()
            B = tr = None
        except Exception:  # noqa: BLE001
            import traceback
            errs.append(traceback.format_exc())
            if hub is not None:
                capi.load().gvt_hip_hub_abort(hub)
        finally:
            import gc
            gc.collect()
            if ctx is not None:
                ctx.close()

    if world == 1:
        rank_main(0)
    else:
        th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
        [t.start() for t in th]
        [t.join(timeout=600) for t in th]
        capi.load().gvt_hip_hub_destroy(hub)
    assert not errs, errs[0]
    got = frames[0]
    for k, (fb, ref) in enumerate(zip(got, refs)):
        tol = 1e-5 if path[k].depth > 1 else 0.0
        assert np.abs(fb[..., :3] - ref[..., :3]).max() <= tol, k
        assert np.array_equal(fb[..., 3], ref[..., 3]), k
    assert np.array_equal(got[3], got[0]) and np.array_equal(got[7], got[0])


def test_soup_domains_with_cross_traffic_native(hip):
    sc = scenes.soup_domains_scene(200_000, 4, 320, 180)
    sc.camera.eye, sc.camera.focus = (3.0, 0.6, 0.4), (0.5, 0.5, 0.5)  # along -x: rays cross the x-tiled domains
    sc.lights["position"] = (2.0, 2.5, 1.5)
    owner = [i % 2 for i in range(sc.n_inst)]
    for bsp in (True, False):
        res = run_native_ranks(sc, owner, 2, NORMALS_FLAT, bsp)
        ref, st = oracle_render_domain(sc, owner, 2, 0)
        assert np.array_equal(res[0][0][..., :3], ref[..., :3])
        assert sum(r[1]["rays_sent"] for r in res.values()) == st.rays_sent and st.rays_sent > 1000


@pytest.mark.parametrize("case", ["soup2", "soup4", "soup8", "config4", "config5", "config5_8"])
def test_known_miss_shortcut_against_the_reference_rule(hip, case):
    """shuffleRays' known-miss shortcut (gvt_device.h; knob skip_known, OFF by default): a ray is not traced again, nor sent again, in an instance
    it has already crossed without a hit on the same straight segment.  In both modes the device's image and ray counts are the checker's for
    the same rule (the reference's hop-by-hop rule by default; the restated shortcut).  Between the modes: fewer rays traced and sent, fewer
    exchanges -- and the same image EXCEPT where a re-trace from the advanced origin flips an edge-grazing triangle test that missed from the
    earlier origin: the reference then finds a hit (and its shadow rays deposit) that the shortcut never looks for.  Rare (config5_8: 2 of
    147,456 pixels; none in the other cases), which is why the shortcut is an opt-in approximation and not the default."""
    sc, mode, world, exact = {
        "soup2": lambda: (scenes.soup_domains_scene(400_000, 2, 480, 270), NORMALS_FLAT, 2, True),
        "soup4": lambda: (scenes.soup_domains_scene(400_000, 4, 480, 270), NORMALS_FLAT, 4, True),
        "soup8": lambda: (scenes.soup_domains_scene(400_000, 8, 480, 270), NORMALS_FLAT, 8, True),
        "config4": lambda: (scenes.bunny_grid_scene(width=475, height=270), NORMALS_SMOOTH, 8, True),
        "config5": lambda: (config5(256, 4), NORMALS_FLAT, 4, False),
        "config5_8": lambda: (config5(384, 8), NORMALS_FLAT, 8, False)}[case]()
    owner = [i % world for i in range(sc.n_inst)]
    got = {}
    for skip in (0, 1):
        ref, st = oracle_render_domain(sc, owner, world, mode, rule="shortcut" if skip else "strict")
        for bsp in (True, False):
            res = run_native_ranks(sc, owner, world, mode, bsp, opts=(("skip_known", skip),))
            fb = res[0][0]
            if exact:
                assert np.array_equal(fb[..., :3], ref[..., :3])
            else:
                assert np.abs(fb[..., :3] - ref[..., :3]).max() <= 1e-5
            assert np.array_equal(fb[..., 3], ref[..., 3])
            assert sum(r[1]["rays_sent"] for r in res.values()) == st.rays_sent
            assert sum(r[1]["rays_closest"] for r in res.values()) == st.rays_closest and sum(r[1]["rays_any"] for r in res.values()) == st.rays_any
            if bsp:
                assert all(r[1]["rounds"] == st.rounds for r in res.values())
            got[(skip, bsp)] = (fb, max(r[1]["rounds"] for r in res.values()), st.rays_sent, st.rays_closest)
    for bsp in (True, False):
        a, b = got[(0, bsp)], got[(1, bsp)]
        flipped = a[0][..., 3] != b[0][..., 3]  # pixels whose deposit COUNT differs: an edge-grazing flip of the reference's re-trace
        assert flipped.sum() <= max(2, flipped.size // 20000)
        if case != "config5_8":
            assert not flipped.any()
        same = ~flipped
        assert np.array_equal(a[0][same], b[0][same]) if exact else np.abs(a[0][same] - b[0][same]).max() <= 1e-5
        assert b[1] <= a[1] and b[2] <= a[2] and b[3] <= a[3]
    if case.startswith("soup"):  # the tiles' boxes overlap: the hand-back hops are there to be skipped
        assert got[(1, True)][1] < got[(0, True)][1] and got[(1, True)][2] < got[(0, True)][2]


def _toy_crossing_scene():
    sc = scenes.soup_domains_scene(30000, 4, 96, 54)
    sc.camera.eye, sc.camera.focus = (3.0, 0.6, 0.4), (0.5, 0.5, 0.5)  # along -x: rays cross the x-tiled domains one after the other
    sc.lights["position"] = (2.0, 2.5, 1.5)
    return sc


@pytest.mark.parametrize("world,bsp", [(2, False), (2, True), (4, False), (3, True)])
def test_small_payloads_ride_inside_the_announce(hip, world, bsp):
    """inline_kb: a pair's payload of a tick that fits the inline area travels INSIDE the announce message -- SendRays' count exchange and ray
    exchange (DomainTracer.h:397-415, :433-463) in ONE transport group per tick.  Same image, same rays crossing the same boundaries as with the
    two-step exchange (inline_kb = 0) and as the restated DomainTracer; with a 1 KiB area a frame mixes both routes."""
    sc = _toy_crossing_scene()
    owner = [i % world for i in range(sc.n_inst)]
    ref, st = oracle_render_domain(sc, owner, world, 0)
    res = {kb: run_native_ranks(sc, owner, world, NORMALS_FLAT, bsp, opts=(("inline_kb", kb),)) for kb in (0, 1, 1024)}
    for kb, r in res.items():
        assert np.array_equal(r[0][0][..., :3], ref[..., :3]), kb
        assert sum(x[1]["rays_sent"] for x in r.values()) == st.rays_sent and st.rays_sent > 100
        assert sum(x[1]["rays_closest"] for x in r.values()) == st.rays_closest
    sent = lambda kb, k: sum(x[1][k] for x in res[kb].values())  # noqa: E731
    assert sent(0, "rays_inline") == 0
    assert 0 < sent(1, "rays_inline") < sent(1, "rays_sent")           # pairs with at most 12 rays inline, the others behind the announce
    assert sent(1024, "rays_inline") == sent(1024, "rays_sent")        # everything fits
    for x in res[1024].values():                                         # ONE group per tick (+ at most one for the composite)
        assert x[1]["rounds"] <= x[1]["exchanges"] <= x[1]["rounds"] + 1
    assert all(x[1]["exchanges"] > x[1]["rounds"] + 1 for x in res[0].values())
    if not bsp:  # what has arrived with the announce is traced in the very next tick: never more ticks than the two-step exchange needs
        assert max(x[1]["rounds"] for x in res[1024].values()) <= max(x[1]["rounds"] for x in res[0].values())


@pytest.mark.parametrize("world", [2, 3, 4])
def test_speculative_tick_parts_change_nothing_but_the_timing(hip, world):
    """spec_ticks (default on): behind every announce exchange of an asynchronous Domain frame the library enqueues the NEXT tick's small round (k_finish, sized on
    the device from the queues' count words) and its report before the host has read the exchange; k_publish voids them on the device when the exchange calls for
    the host (a payload beyond the inline area in or out, an error, more than finish_rays local rays).  The exchange sequence is untouched, so with the knob on
    or off the same rays cross the same boundaries in the same ticks and the image is the checker's -- also with an inline area so small that most parts are void,
    and with finish_rays so small that every part is."""
    sc = _toy_crossing_scene()
    owner = [i % world for i in range(sc.n_inst)]
    ref, st = oracle_render_domain(sc, owner, world, 0)
    base = None
    for opts in ((("spec_ticks", 0),), (("spec_ticks", 1),), (("spec_ticks", 1), ("inline_kb", 1)), (("spec_ticks", 1), ("finish_rays", 16)), (("spec_ticks", 1), ("inline_kb", 1024))):
        r = run_native_ranks(sc, owner, world, NORMALS_FLAT, False, opts=opts)
        assert np.array_equal(r[0][0][..., :3], ref[..., :3]) and np.array_equal(r[0][0][..., 3], ref[..., 3]), opts
        got = (sum(x[1]["rays_sent"] for x in r.values()), sum(x[1]["rays_closest"] for x in r.values()), sum(x[1]["rays_any"] for x in r.values()))
        assert got == (st.rays_sent, st.rays_closest, st.rays_any), opts
        shape = ([x[1]["rounds"] for x in r.values()], [x[1]["exchanges"] for x in r.values()])
        if dict(opts).get("inline_kb", 16) == 16 and "finish_rays" not in dict(opts):
            base = base or shape
            assert shape == base, (opts, shape, base)  # same ticks, same transport groups: the peers cannot tell


@pytest.mark.parametrize("world,bsp", [(2, False), (3, False), (2, True)])
def test_exchanges_on_the_communicators_own_stream(hip, world, bsp):
    """comm_stream = 1 (GVT_HIP_COMM_STREAM): the announce, k_publish's inline appends and the payloads run on the communicator's own stream; the compute
    stream waits for the WHOLE of k_publish (not only for the block that releases the host's sequence word) before the next chain reads the queues
    (an all-inline tick used to leave that unordered).  Same image and counts as the restated DomainTracer, frame after frame."""
    sc = _toy_crossing_scene()
    owner = [i % world for i in range(sc.n_inst)]
    ref, st = oracle_render_domain(sc, owner, world, 0)
    for kb in (1024, 16, 0):
        for _ in range(3):
            r = run_native_ranks(sc, owner, world, NORMALS_FLAT, bsp, opts=(("comm_stream", 1), ("inline_kb", kb)))
            assert np.array_equal(r[0][0][..., :3], ref[..., :3]) and np.array_equal(r[0][0][..., 3], ref[..., 3]), kb
            assert sum(x[1]["rays_sent"] for x in r.values()) == st.rays_sent
            assert sum(x[1]["rays_closest"] for x in r.values()) == st.rays_closest and sum(x[1]["rays_any"] for x in r.values()) == st.rays_any


def test_reserved_compute_units_for_the_communicators_stream(hip):
    """comm_cus: the communicator's own stream gets k compute units to itself (CU-masked stream), the rank's compute stream the others, the
    persistent grids are sized for those; with payload_overlap_kb = 0 and inline_kb = 0 every payload moves on that stream beside the next
    chain.  Same image and counts as without the reservation."""
    sc = config5(256, 4)
    owner = [i % 2 for i in range(sc.n_inst)]
    ref, st = oracle_render_domain(sc, owner, 2, 0)
    for opts in ((("comm_cus", 16), ("payload_overlap_kb", 0), ("inline_kb", 0)), (("comm_cus", 8),)):
        res = run_native_ranks(sc, owner, 2, NORMALS_FLAT, False, opts=opts)
        assert np.abs(res[0][0][..., :3] - ref[..., :3]).max() <= 1e-5 and np.array_equal(res[0][0][..., 3], ref[..., 3])
        assert sum(r[1]["rays_sent"] for r in res.values()) == st.rays_sent
        assert all(r[1]["reserved_cus"] == dict(opts)["comm_cus"] for r in res.values())


def test_composite_rectangles_equal_the_full_reduce(hip):
    """The composite sends rank 0 each rank's written rectangle (known from the last announce); the result equals the sum-reduce of
    whole frames, on a scene where ranks write overlapping pixel sets (bounce rays deposit anywhere)."""
    sc = config5(256, 4)
    owner = [0, 1, 2, 1]
    a = run_native_ranks(sc, owner, 3, NORMALS_FLAT, False)[0][0]
    b = run_native_ranks(sc, owner, 3, NORMALS_FLAT, False, full_reduce=True)[0][0]
    assert np.abs(a - b).max() <= 1e-5 and np.array_equal(a[..., 3], b[..., 3]) and (a[..., 3] > 0).mean() > 0.2


@pytest.mark.parametrize("world", [2, 3, 4])
def test_image_scheduler_on_several_ranks(hip, world):
    """Tracer<ImageScheduler> under several ranks (ImageTracer.h:111-125): the scene replicated, each rank traces its contiguous
    portion of the camera's rays, no ray changes rank, composite on rank 0.  The image is the one-rank image; the ranks' ray
    counts add up to the one-rank counts."""
    for sc, mode, tol in ((scenes.bunny_grid_scene(width=380, height=216), NORMALS_SMOOTH, 0.0), (config5(192, 4), NORMALS_FLAT, 1e-5)):
        res = run_native_ranks(sc, [0] * sc.n_inst, world, mode, False, image=True)
        ref, st = oracle_render(sc, mode, nthreads=8)
        fb = res[0][0]
        assert np.abs(fb[..., :3] - ref[..., :3]).max() <= tol and np.array_equal(fb[..., 3], ref[..., 3])
        assert sum(r[1]["rays_closest"] for r in res.values()) == st.rays_closest and sum(r[1]["rays_any"] for r in res.values()) == st.rays_any
        assert all(r[1]["rays_sent"] == 0 for r in res.values())  # (a portion of the film may see nothing at all)


def test_a_failing_rank_ends_the_frame_on_every_rank_and_the_next_frame_works(hip):
    """A rank whose local work fails (test knob: at its second exchange) still joins the announce with its error word: it leaves with
    its own error, every other rank with GVT_HIP_ERR_PEER naming it -- nobody hangs -- and the ranks are still in step: the next
    frame of the same tracers gives the oracle's image."""
    world = 3
    sc = config5(192, 6)
    owner = [i % world for i in range(sc.n_inst)]
    hub = capi.load().gvt_hip_hub_create(world)
    msgs, fbs, errs = {}, {}, []
    sync = threading.Barrier(world)

    def rank_main(rank):
        ctx = None
        try:
            ctx = Context(0)
            comm = Comm.local(hub, rank)
            comm.set_deadline_ms(20000)
            assert comm.count == world
            tr = NativeTracer(sc, NORMALS_FLAT, owner, comm)
            if rank == 1:
                capi.set_option("inject_fail_tick", 1)
            try:
                tr()
                msgs[rank] = "no error"
            except capi.GvtHipError as e:
                msgs[rank] = str(e)
            capi.set_option("defaults", 0)
            sync.wait(timeout=60)
            B = tr()
            fbs[rank] = B.framebuffer(True) if rank == 0 else None
            tr.close(); comm.close()
            B = tr = None
        except Exception:  # noqa: BLE001
            import traceback
            errs.append(traceback.format_exc())
            capi.load().gvt_hip_hub_abort(hub)
        finally:
            import gc
            gc.collect()
            if ctx is not None:
                ctx.close()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    capi.load().gvt_hip_hub_destroy(hub)
    assert not errs, errs[0]
    assert "injected failure" in msgs[1], msgs
    assert all("rank 1 reported error" in msgs[r] for r in (0, 2)), msgs
    ref, _ = oracle_render_domain(sc, owner, world, 0)
    assert np.abs(fbs[0][..., :3] - ref[..., :3]).max() <= 1e-5 and np.array_equal(fbs[0][..., 3], ref[..., 3])


def test_ranks_that_disagree_on_the_message_layout_are_told_so(hip):
    """inline_kb differs between two ranks: the announce messages would have different lengths (RCCL: a hang or a corrupt row).  The frame's first exchange
    is a 16-byte layout handshake; both ranks come back with GVT_HIP_ERR_INVALID naming the knob, nobody waits for a deadline; with the knob
    agreed again the same tracers render the oracle's image."""
    sc = _toy_crossing_scene()
    owner = [i % 2 for i in range(sc.n_inst)]
    hub = capi.load().gvt_hip_hub_create(2)
    msgs, fbs, errs = {}, {}, []
    sync = threading.Barrier(2)

    def rank_main(rank):
        ctx = None
        try:
            ctx = Context(0)
            capi.set_option("inline_kb", 16 if rank == 0 else 4)
            comm = Comm.local(hub, rank)
            tr = NativeTracer(sc, NORMALS_FLAT, owner, comm)
            try:
                tr()
                msgs[rank] = "no error"
            except capi.GvtHipError as e:
                msgs[rank] = str(e)
            capi.set_option("inline_kb", 16)
            sync.wait(timeout=60)
            B = tr()
            fbs[rank] = B.framebuffer(True) if rank == 0 else None
            tr.close(); comm.close()
            B = tr = None
        except Exception:  # noqa: BLE001
            import traceback
            errs.append(traceback.format_exc())
            capi.load().gvt_hip_hub_abort(hub)
        finally:
            import gc
            gc.collect()
            if ctx is not None:
                ctx.close()

    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(2)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    capi.load().gvt_hip_hub_destroy(hub)
    assert not errs, errs[0]
    assert all("laid out differently" in msgs[r] and "inline_kb" in msgs[r] for r in (0, 1)), msgs
    ref, _ = oracle_render_domain(sc, owner, 2, 0)
    assert np.array_equal(fbs[0][..., :3], ref[..., :3])


def test_a_rank_that_never_joins_runs_its_peer_into_the_deadline(hip):
    """Rank 1 never calls the frame: rank 0 must come back from its first announce with GVT_HIP_ERR_TIMEOUT after the communicator's
    deadline (here 400 ms), not hang."""
    import time

    sc = scenes.bunny_grid_scene(width=190, height=108)
    owner = [i % 2 for i in range(sc.n_inst)]
    hub = capi.load().gvt_hip_hub_create(2)
    res = {}

    def rank0():
        ctx = Context(0)
        try:
            comm = Comm.local(hub, 0)
            comm.set_deadline_ms(400)
            tr = NativeTracer(sc, NORMALS_SMOOTH, owner, comm)
            t0 = time.perf_counter()
            try:
                tr()
                res["msg"] = "no error"
            except capi.GvtHipError as e:
                res["msg"] = str(e)
            res["s"] = time.perf_counter() - t0
            tr.close(); comm.close()
            tr = None
        finally:
            import gc
            gc.collect()
            ctx.close()

    t = threading.Thread(target=rank0)
    t.start()
    t.join(timeout=120)
    assert not t.is_alive(), "rank 0 hangs in an exchange its peer never joined"
    capi.load().gvt_hip_hub_destroy(hub)
    assert "waited more than 400 ms" in res["msg"] and "a peer has stopped taking part" in res["msg"], res
    assert res["s"] < 30.0


def test_rccl_communicator_of_one_rank(hip):
    """RCCL itself, as far as one GPU allows: the library resolves librccl, creates a communicator from a unique id and runs a
    Domain frame on it (no peers: no sends; the composite reduce is skipped for one rank)."""
    uid = Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = Comm.rccl(uid, 0, 1)
    assert comm.rank == 0 and comm.world == 1 and comm.count == 1  # ncclCommCount agrees with the world the launcher asked for
    comm.selftest(1 << 20)  # ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd to self + ncclReduce, on the communication stream
    comm.selftest(80 * 1000 + 8 * 3)  # a payload-shaped size (three queues' headers + 1000 rays)
    sc = scenes.bunny_grid_scene(width=190, height=108)
    tr = NativeTracer(sc, NORMALS_SMOOTH, [0] * sc.n_inst, comm)
    fb = tr(bsp=True).framebuffer(True)
    ref, _ = oracle_render(sc, 1)
    assert np.array_equal(fb[..., :3], ref[..., :3])
    tr.close()
    comm.close()


def test_two_contexts_trace_concurrently(hip):
    """Per-context state: two threads, each with its own context, render different scenes at the same time; both equal the oracle."""
    scs = [scenes.bunny_grid_scene(width=240, height=136), scenes.simple_scene(192, 192)]
    refs = [oracle_render(s, 1)[0] for s in scs]
    out, errs = {}, []

    def work(k):
        try:
            with Context(0):
                for _ in range(3):
                    out[k] = ImageTracer(scs[k], NORMALS_SMOOTH)().framebuffer(True)
        except Exception:  # noqa: BLE001
            import traceback
            errs.append(traceback.format_exc())

    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in th]
    [t.join(timeout=300) for t in th]
    assert not errs, errs[0]
    for k in range(2):
        assert np.array_equal(out[k][..., :3], refs[k][..., :3])


def many_instances_scene(nx, ny, width, height):
    """nx * ny instances of two small meshes (cube / cone like gvtSimple) on a jittered grid with overlapping boxes."""
    base = scenes.simple_scene(width, height)
    rng = np.random.Generator(np.random.Philox(5))
    mats, inst_mesh = [], []
    for j in range(ny):
        for i in range(nx):
            s = 0.35 + 0.3 * rng.random()
            t = ((i - nx / 2) * 0.9 + 0.3 * rng.random(), (j - ny / 2) * 0.9 + 0.3 * rng.random(), 2.0 * rng.random())
            mats.append(scenes.mat_translate_scale(t, (s, s, s)))
            inst_mesh.append((i + j) % len(base.meshes))
    cam = scenes.Camera((0.0, 0.0, 40.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 0.9, width, height, 1, 1, 0.0)
    from gravit_amd.layouts import point_light
    return scenes._assemble(base.meshes, inst_mesh, mats, point_light((5.0, 8.0, 30.0)), cam, "grid-%dx%d" % (nx, ny))


def test_top_level_bvh_with_1056_instances(hip):
    """A top-level set far beyond the linear-scan and scan-ordered-shuffle sizes (1,056 instances): the device walks the
    reference BVH's own nodes (left child first, RayPacket's node test) and must pick the instance the oracle's leaf-order scan
    picks, for camera rays and for every moved ray of the frame -- image, ray counts and adapter-call count equal the oracle."""
    from gravit_amd.adapter import RayQueue, TopLevel, camera_generate
    from oracle import orc
    from tests.helpers import bits, oracle_camera_rays

    sc = many_instances_scene(33, 32, 384, 384)
    assert sc.n_inst == 1056
    top = TopLevel(sc.inst_lo, sc.inst_hi)
    assert (top.order() == orc.toplevel_order(sc.inst_lo, sc.inst_hi)).all()
    # per-ray decisions of FilterRaysLocally against the oracle's scan
    rays = oracle_camera_rays(sc)
    q = RayQueue()
    camera_generate(q, sc.camera, 0)
    queues = [RayQueue() for _ in range(sc.n_inst)]
    top.shuffle(q, -1, queues, None)
    nxt, t = orc.toplevel_intersect(sc.inst_lo, sc.inst_hi, top.order(), rays, -1)
    sizes = np.array([len(x) for x in queues])
    assert (sizes == np.bincount(nxt[nxt >= 0], minlength=sc.n_inst)).all() and (sizes > 0).sum() > 500
    for i in np.nonzero(sizes)[0][:40]:
        got = queues[i].to_numpy()
        exp = rays[nxt == i].copy()
        exp["origin"] = exp["origin"] + exp["direction"] * (t[nxt == i] * np.float32(0.95))[:, None]
        assert (np.sort(got["id"]) == np.sort(exp["id"])).all()
        o = np.argsort(got["id"]); e = np.argsort(exp["id"])
        assert (bits(got["origin"][o]) == bits(exp["origin"][e])).all()
    # whole frames: rounds and the reference-order loop
    ref, st = oracle_render(sc, NORMALS_SMOOTH, nthreads=8)
    assert (ref[..., :3].sum(axis=2) > 0).mean() > 0.1 and st.adapter_calls > 1000
    tr = NativeTracer(sc, NORMALS_SMOOTH)
    fb = tr().framebuffer(True)
    assert np.array_equal(fb[..., :3], ref[..., :3]) and tr.stats["rays_closest"] == st.rays_closest and tr.stats["rays_any"] == st.rays_any
    it = ImageTracer(sc, NORMALS_SMOOTH)
    assert np.array_equal(it().framebuffer(True)[..., :3], ref[..., :3]) and it.adapter_calls == st.adapter_calls


@pytest.mark.parametrize("opts", [dict(small_rays=0), dict(small_rays=1 << 30), dict(term_sink=0), dict(leaf_max=4, small_rays=0), dict(finish_rays=0), dict(round_room_mb=0),
                                  dict(round_room_mb=0, finish_rays=0, small_rays=0), dict(finish_rays=1 << 30), dict(finish_rays=1 << 30, leaf_max=4), dict(skip_known=1, finish_rays=0), dict(skip_known=1),
                                  dict(sort_rays=1), dict(packet=0), dict(packet=2), dict(packet=2, long_steps=0), dict(packet=2, term_sink=0), dict(packet=1, packet_min_rays=0),
                                  dict(finish_clusters=0), dict(finish_clusters=0, finish_rays=1 << 30), dict(hop_local=0), dict(hop_local=2), dict(hop_local=3), dict(hop_local=2, term_sink=0), dict(hop_local=2, skip_known=1), dict(shadow_order=0), dict(shadow_order=1, shadow_order_min_rays=0, long_min_rays=0), dict(shadow_order=1, shadow_order_min_rays=0, long_min_rays=0, packet=0, term_sink=0)])
def test_round_results_do_not_depend_on_knobs(hip, opts):
    """The round chain under every knob of the shipped library -- a wave per ray for small rounds or never, k_finish or rounds (over the cluster layout of the
    nodes, the default, or over the plain 4-wide nodes), exact
    growth, no terminal sink, the known-miss shortcut (against the checker's restatement of it) -- returns the oracle's image on a multi-domain depth-2 frame, on config 4
    and on a soup (the variants that lost -- merged kernels for one queue, compacted shadow slots, non-lean frames, k_fused / k_packet /
    k_traceq -- tests/experiment_cases.py, against the experiments build)."""
    for sc, mode, tol in ((config5(192, 4), NORMALS_FLAT, 1e-5), (scenes.bunny_grid_scene(width=380, height=216), NORMALS_SMOOTH, 0.0),
                          (scenes.soup_scene(100_000, 160, 90), NORMALS_FLAT, 0.0)):
        ref, st = oracle_render(sc, mode, nthreads=8, rule="shortcut" if opts.get("skip_known", 0) else "strict")
        try:
            for k, v in opts.items():
                hip.set_option(k, v)
            tr = NativeTracer(sc, mode)
            fb = tr().framebuffer(True)
            assert np.abs(fb[..., :3] - ref[..., :3]).max() <= tol and np.array_equal(fb[..., 3], ref[..., 3])
            assert tr.stats["rays_closest"] == st.rays_closest and tr.stats["rays_any"] == st.rays_any
            tr.close()
        finally:
            hip.set_option("defaults", 0)


def test_sorted_bounce_lists_change_nothing_but_the_order(hip):
    """sort_rays in the native chain (round 6): in a single-mesh round with bounces the list of secondary rays is reordered by direction octant and the Morton cell
    of the ray's entry point before the closest-hit launch (its length lives on the device: the whole bound is sorted, unused entries last), the pass's shadow rays
    inherit the order.  A ray's RNG stream travels with the ray, so the one-instance hall at depth 3 gives the oracle's image and ray counts with the knob on."""
    sc = scenes.cathedral_scene(160, 160, samples=2, depth=3, eye=(0.0, 1.5, 13.0), light=(0.0, 2.5, 12.0))
    assert sc.n_inst == 1
    ref, st = oracle_render(sc, NORMALS_FLAT, nthreads=8)
    try:
        for on in (0, 1):
            hip.set_option("sort_rays", on)
            hip.stats_reset(); hip.profile(1)
            tr = NativeTracer(sc, NORMALS_FLAT)
            fb = tr().framebuffer(True)
            sorted_ms = hip.stats()["ms_sort"]
            hip.profile(False)
            assert np.abs(fb[..., :3] - ref[..., :3]).max() <= 1e-5 and np.array_equal(fb[..., 3], ref[..., 3]), on
            assert tr.stats["rays_closest"] == st.rays_closest > 100_000 and tr.stats["rays_any"] == st.rays_any, on
            assert (sorted_ms > 0.0) == bool(on)  # the sort really ran (two bounce passes) / never ran
            tr.close()
    finally:
        hip.set_option("defaults", 0)


def test_shadow_rays_are_listed_by_their_primaries_step_counts(hip):
    """knob shadow_order: a single-mesh round with one light writes its shadow rays into eight regions, by the node steps the longest primary of their 64-ray
    tile took (the closest-hit launch leaves a byte per ray), and the any-hit launch walks the regions from the longest class down.  Same image, same ray
    counts, with the knob on and off; the regions' counts add up to the shadow rays traced; depth 2 (a second pass over an index list) and a mesh with
    packets (no per-ray step counts: the knob does not apply) likewise."""
    for sc, mode, depth2 in ((scenes.soup_scene(100_000, 320, 180), NORMALS_FLAT, False), (scenes.soup_scene(100_000, 160, 90), NORMALS_FLAT, True), (scenes.bunny_scene(256, 256), NORMALS_SMOOTH, False)):
        if depth2:
            from dataclasses import replace
            sc = replace(sc, camera=replace(sc.camera, depth=2))
        ref, st = oracle_render(sc, mode, nthreads=8)
        imgs = {}
        for on in (0, 1):
            try:
                hip.set_option("shadow_order", on); hip.set_option("shadow_order_min_rays", 0); hip.set_option("long_min_rays", 0); hip.set_option("packet_min_rays", 0)
                tr = NativeTracer(sc, mode)
                imgs[on] = tr().framebuffer(True).copy()
                assert tr.stats["rays_closest"] == st.rays_closest and tr.stats["rays_any"] == st.rays_any
                classes = hip.counters_peek()[24:32]
                packets = tr.backend.adapter(0).info()["packet"]
                if on and not packets and not depth2:
                    assert sum(classes) == st.rays_any and sum(1 for c in classes if c) >= 2, classes
                if not on or packets:
                    assert not any(classes), classes
                tr.close()
            finally:
                hip.set_option("defaults", 0)
        tol = 1e-5 if depth2 else 0.0
        assert np.abs(imgs[1][..., :3] - ref[..., :3]).max() <= tol and np.array_equal(imgs[1][..., 3], ref[..., 3])
        assert np.abs(imgs[0] - imgs[1]).max() <= tol


@pytest.mark.parametrize("mode", [NORMALS_FLAT, NORMALS_SMOOTH])
def test_the_lean_shading_kernel_and_its_conditions(hip, mode):
    """k_shade's LEAN instantiation (csrc/shade.inc: one instance in the scene, depth 1, point / ambient lights, a LAMBERT mesh material) against the oracle --
    one light and three (point, ambient, point), soup and bunny, packets or lanes; and every way OUT of its conditions on the same scene (an area light, depth 2,
    a PHONG material, vertex colours, two instances) through the general kernel: each image equals the oracle's (1e-5 where several deposits per pixel add up in
    any order).  (Its first run found a bug in the BINDING: np.concatenate packs light records to 48 bytes; scheduler.py now always hands over the 64-byte layout.)"""
    from dataclasses import replace

    from gravit_amd import layouts
    base = scenes.soup_scene(60_000, 256, 144)
    bun = scenes.bunny_scene(192, 192)
    three = np.concatenate([layouts.point_light((0.5, 0.5, 3.0)), layouts.ambient_light((0.05, 0.04, 0.03)), layouts.point_light((2.0, 1.5, 2.5), (0.9, 0.8, 0.7))])
    area = np.concatenate([layouts.point_light((0.5, 0.5, 3.0)), layouts.area_light((0.4, 2.0, 1.0), (0.8, 0.8, 1.0), (0.0, -1.0, 0.1), 0.3, 0.2)])
    m0 = base.meshes[0]
    phong = scenes.MeshData(m0.verts, m0.tris, layouts.default_material(kd=(0.6, 0.5, 0.4), mtype=layouts.PHONG, ks=(0.3, 0.3, 0.3), alpha=8.0))
    coloured = scenes.MeshData(m0.verts, m0.tris, m0.material, None, np.random.default_rng(3).random(m0.verts.shape).astype(np.float32))
    cases = [("lean, one light", base, 0.0), ("lean, three lights", replace(base, lights=three), 1e-5), ("lean, bunny", bun, 0.0), ("lean, bunny, three lights", replace(bun, lights=three), 1e-5),
             ("general: an area light", replace(base, lights=area), 1e-5), ("general: depth 2", replace(base, camera=replace(base.camera, depth=2)), 1e-5),
             ("general: PHONG", replace(base, meshes=[phong]), 1e-5), ("general: vertex colours", replace(base, meshes=[coloured]), 0.0),
             ("general: two instances", scenes.soup_domains_scene(60_000, 2, 256, 144), 0.0)]
    for name, sc, tol in cases:
        ref, st = oracle_render(sc, mode, nthreads=8)
        assert (ref[..., 3] > 0).sum() > 500, name
        for pk in (0, 1):
            try:
                hip.set_option("packet", pk); hip.set_option("packet_min_rays", 0)
                tr = NativeTracer(sc, mode)
                fb = tr().framebuffer(True)
                assert np.abs(fb[..., :3] - ref[..., :3]).max() <= tol and np.array_equal(fb[..., 3], ref[..., 3]), (name, pk)
                assert tr.stats["rays_closest"] == st.rays_closest and tr.stats["rays_any"] == st.rays_any, (name, pk)
                tr.close()
            finally:
                hip.set_option("defaults", 0)


def test_packet_traversal_is_chosen_per_mesh(hip):
    """The builder decides per mesh whether coherent lists are traversed a packet of 64 rays per wave (k_packet) or a lane per ray (k_trace), as the
    reference picks its packet width per build (EmbreeMeshAdapter.cpp:50-74): from sum(area(inner node)) / area(root) -- a surface stays at a few
    dozen, a volume-filling soup grows with N^(1/3).  Either way, forced on, forced off: the same bits (one-instance frames: the single-mesh
    kernels; camera rays in tile order and their direct-mapped shadow rays)."""
    from gravit_amd.adapter import HipMeshAdapter

    bunny, soup = scenes.bunny_scene(320, 320), scenes.soup_scene(300_000, 320, 180)
    ib, isoup = HipMeshAdapter(bunny.meshes[0]).info(), HipMeshAdapter(soup.meshes[0]).info()
    assert ib["packet"] == 1 and 1.0 < ib["sah_inner"] < 128.0
    assert isoup["packet"] == 0 and isoup["sah_inner"] > 128.0
    for sc, mode in ((bunny, NORMALS_SMOOTH), (soup, NORMALS_FLAT)):
        ref, st = oracle_render(sc, mode, nthreads=8)
        for pk in (1, 0, 2):
            try:
                hip.set_option("packet", pk)
                hip.set_option("packet_min_rays", 0)  # (the test's film is far below the launch size packets are kept for)
                tr = NativeTracer(sc, mode)
                fb = tr().framebuffer(True)
                assert np.array_equal(fb[..., :3], ref[..., :3]) and np.array_equal(fb[..., 3], ref[..., 3]), (pk,)
                assert tr.stats["rays_closest"] == st.rays_closest and tr.stats["rays_any"] == st.rays_any
                tr.close()
            finally:
                hip.set_option("defaults", 0)


def test_degenerate_scenes_through_the_native_tracer(hip):
    """No lights (nothing is shaded into the frame), an instance whose mesh has no triangles, a camera that sees nothing: frames
    finish, are black where they must be, and a later normal frame of the same process is unaffected."""
    sc = scenes.bunny_grid_scene(width=160, height=90)
    sc.lights = sc.lights[:0]
    tr = NativeTracer(sc, NORMALS_SMOOTH)
    fb = tr().framebuffer(True)
    assert float(np.abs(fb).max()) == 0.0 and tr.stats["rays_closest"] > 1000 and tr.stats["rays_any"] == 0
    tr.close()
    one = scenes.soup_scene(1000, 96, 54)
    one.meshes[0] = scenes.MeshData(np.zeros((3, 3), np.float32), np.zeros((0, 3), np.int32), one.meshes[0].material)
    tr = NativeTracer(one, NORMALS_FLAT)
    assert float(np.abs(tr().framebuffer(True)).max()) == 0.0
    tr.close()
    away = scenes.soup_scene(20_000, 96, 54)
    away.camera.focus = (0.5, 0.5, 9.0)  # looks away from the soup
    tr = NativeTracer(away, NORMALS_FLAT)
    assert float(np.abs(tr().framebuffer(True)).max()) == 0.0 and tr.stats["rays_closest"] == 0
    tr.close()
    ok = scenes.soup_scene(20_000, 96, 54)
    tr = NativeTracer(ok, NORMALS_FLAT)
    ref, _ = oracle_render(ok, NORMALS_FLAT)
    assert np.array_equal(tr().framebuffer(True)[..., :3], ref[..., :3])
    tr.close()


def test_rounds_with_materials_several_lights_and_vertex_colours(hip):
    """The merged kernels take every instance's shading attributes from the per-instance table: per-face Phong / Blinn materials on
    one mesh, vertex colours on another, three lights (point, area: RNG stream per ray, ambient), depth 2, 4 rays per pixel, 6
    instances on one rank and on 3 in-process ranks -- against the oracle's schedulers (1e-5: powf and float-add order)."""
    from gravit_amd import layouts

    base = scenes.bunny_scene(64, 64).meshes[0]
    rng = np.random.default_rng(11)
    mats = np.concatenate([layouts.default_material(kd=rng.random(3), mtype=(layouts.PHONG if k % 2 else layouts.BLINN), ks=rng.random(3),
                                                    alpha=1 + 4 * rng.random()) for k in range(4)])
    face_mat = (np.arange(len(base.tris)) % 5 - 1).astype(np.int32)
    m_a = scenes.MeshData(base.verts, base.tris, layouts.default_material(kd=(0.7, 0.6, 0.5)), None, None, mats, face_mat)
    m_b = scenes.MeshData(base.verts, base.tris, layouts.default_material(), None, rng.random(base.verts.shape).astype(np.float32))
    grid = scenes.bunny_grid_scene(nx=3, ny=2, pitch=0.25, width=300, height=200)
    lights = np.concatenate([layouts.point_light((0.0, 0.4, 1.0)), layouts.area_light((0.3, 0.9, 0.6), (0.8, 0.8, 1.0), (0.0, -1.0, 0.1), 0.3, 0.2),
                             layouts.ambient_light((0.03, 0.03, 0.05))])
    sc = scenes._assemble([m_a, m_b], [i % 2 for i in range(grid.n_inst)], [grid.m[i].reshape(16) for i in range(grid.n_inst)], lights, grid.camera, "materials-grid")
    sc.camera.samples, sc.camera.depth = 2, 2
    ref, st = oracle_render(sc, NORMALS_SMOOTH, nthreads=8)
    assert (ref[..., :3].sum(axis=2) > 0).mean() > 0.02 and st.rays_any > 10_000
    tr = NativeTracer(sc, NORMALS_SMOOTH)
    fb = tr().framebuffer(True)
    assert np.abs(fb[..., :3] - ref[..., :3]).max() <= 1e-5 and np.array_equal(fb[..., 3], ref[..., 3])
    assert tr.stats["rays_closest"] == st.rays_closest and tr.stats["rays_any"] == st.rays_any
    tr.close()
    owner = [i % 3 for i in range(sc.n_inst)]
    res = run_native_ranks(sc, owner, 3, NORMALS_SMOOTH, False)
    refd, std = oracle_render_domain(sc, owner, 3, 1)
    assert np.abs(res[0][0][..., :3] - refd[..., :3]).max() <= 1e-5 and np.array_equal(res[0][0][..., 3], refd[..., 3])
    assert sum(r[1]["rays_sent"] for r in res.values()) == std.rays_sent


def test_tracer_contexts_and_queues_release_their_memory(hip):
    """Create / render / destroy tracers, communicators and contexts repeatedly: free device memory returns to where it was
    (scratch arenas of the default context are grow-only and warmed up first)."""
    import ctypes
    import gc

    rt = ctypes.CDLL("libamdhip64.so")
    def free_bytes():
        hip.synchronize()
        f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert rt.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
        return f.value

    sc = scenes.bunny_grid_scene(width=240, height=136)
    def cycle():
        tr = NativeTracer(sc, NORMALS_SMOOTH)
        tr(); tr()
        tr.close()
        res = run_native_ranks(sc, [i % 2 for i in range(sc.n_inst)], 2, NORMALS_SMOOTH, False)
        assert res[0][0] is not None
        gc.collect()
    cycle(); cycle()
    free0 = free_bytes()
    for _ in range(6):
        cycle()
    free1 = free_bytes()
    assert free0 - free1 < 64 << 20, "leaked %.1f MiB over 6 create/destroy cycles" % ((free0 - free1) / 2 ** 20)


def test_parking_threshold_adapts_to_a_sparser_scene(hip):
    """long_auto (default on): on a scene that parks more than 0.3 % of its closest-hit rays at the benchmark's threshold -- a 1 M-triangle
    soup is sparser than the 10 M one, its rays take more node steps -- the native tracer raises the threshold from frame to frame; the
    frames stay the same bit for bit (parking never changes a result) and far fewer rays end up in k_long_closest."""
    sc = scenes.soup_scene(1_000_000, 960, 540)
    tr = NativeTracer(sc, NORMALS_FLAT)
    fb0 = tr().framebuffer(False).copy()
    parked0, rays = int(hip.counters_peek()[3]), tr.stats["rays_closest"]
    for _ in range(8):
        fb = tr().framebuffer(False)
        assert np.array_equal(fb, fb0)
    parked = int(hip.counters_peek()[3])
    assert rays > 200_000 and parked0 > 0.003 * rays, (parked0, rays)  # the premise: this scene does park a lot at 96 steps
    assert parked < 0.004 * rays and parked < parked0 // 3, (parked0, parked, rays)
    tr.close()
    hip.set_option("long_auto", 0)
    tr = NativeTracer(sc, NORMALS_FLAT)
    fb = tr().framebuffer(False)
    assert np.array_equal(fb, fb0) and int(hip.counters_peek()[3]) == parked0  # off: the knob's threshold, frame after frame
    tr.close()


def test_small_rounds_through_one_launch_or_per_hop_whichever_the_tracer_times_faster(hip):
    """finish_auto (default on; one rank, several instances): frames 2.. alternate between k_finish (a small round followed to its end in
    one launch) and per-hop merged chains until each has been timed three times, then the tracer keeps the faster.  Every frame -- probing
    or settled, either route -- is the oracle's image bit for bit with the oracle's ray counts.  On the soup tiles the routes differ in
    their number of launch chains (rays hop several times), which shows that the probing frames took both and that the choice settled.
    (hop_local = 0 here: the third and fourth route, hops inside the merged launches, have their own test below.)"""
    for sc, mode in ((scenes.bunny_grid_scene(width=760, height=432), NORMALS_SMOOTH), (scenes.soup_domains_scene(200_000, 4, 480, 270), NORMALS_FLAT)):
        ref, st = oracle_render(sc, mode, nthreads=8)
        pinned = {}
        for fin in (32768, 0):  # the two routes pinned
            hip.set_option("finish_auto", 0); hip.set_option("finish_rays", fin); hip.set_option("hop_local", 0)
            try:
                tr = NativeTracer(sc, mode)
                fb = tr().framebuffer(True)
                assert np.array_equal(fb[..., :3].view(np.uint32), ref[..., :3].view(np.uint32))
                assert tr.stats["rays_closest"] == st.rays_closest and tr.stats["rays_any"] == st.rays_any
                pinned[fin] = tr.stats["chains"]
                tr.close()
            finally:
                hip.set_option("defaults", 0)
        chains = []
        try:
            hip.set_option("hop_local", 0)
            tr = NativeTracer(sc, mode)
            for _ in range(12):
                fb = tr().framebuffer(True)
                assert np.array_equal(fb[..., :3].view(np.uint32), ref[..., :3].view(np.uint32))
                assert tr.stats["rays_closest"] == st.rays_closest and tr.stats["rays_any"] == st.rays_any
                chains.append(tr.stats["chains"])
            tr.close()
        finally:
            hip.set_option("defaults", 0)
        assert set(chains) <= set(pinned.values())
        if pinned[0] != pinned[32768]:
            assert chains[:2] == [pinned[32768]] * 2 and set(chains[2:8]) == set(pinned.values()), (chains, pinned)  # probing: both routes
            assert len(set(chains[8:])) == 1, chains  # settled
    assert pinned[0] > pinned[32768]  # (the soup tiles: per-hop rounds need more chains)


def test_hops_into_the_next_local_instance_change_nothing_but_the_rounds(hip):
    """hop_local (round 6): in a merged launch a ray that leaves its instance without a hit and has another instance of THIS rank ahead goes on there inside the
    launch -- the lane applies shuffleRays' rule (origin advanced by 95 % of the entry distance, TracerBase.h:392-400) and starts again in the next instance --
    instead of waiting in that instance's queue for the next round; an un-occluded shadow ray likewise.  Never (0), always (2) or early (3: the closest-hit launch
    only, before its drain), on one rank and on two and three
    (only instances the rank owns are entered), with rays parked for a wave after a few steps (they keep their new instance) and through bounces (a bounce starts
    where its parent was hit): the oracle's image, the oracle's ray counts, the same rays sent; on one rank fewer launch chains.  hop_local = 1 (default): the
    tracer times the routes like finish_auto and every frame, whichever route it took, is the same image."""
    cases = ((scenes.bunny_grid_scene(width=380, height=216), NORMALS_SMOOTH, 0.0), (config5(192, 4), NORMALS_FLAT, 1e-5), (scenes.soup_domains_scene(100_000, 4, 240, 136), NORMALS_FLAT, 0.0))
    for sc, mode, tol in cases:
        ref, st = oracle_render(sc, mode, nthreads=8)
        chains = {}
        for opts in (dict(hop_local=0), dict(hop_local=2), dict(hop_local=3), dict(hop_local=2, finish_rays=0), dict(hop_local=3, finish_rays=0), dict(hop_local=2, long_steps=6, long_min_rays=0), dict(hop_local=2, small_rays=0, packet=0)):
            try:
                for k, v in opts.items():
                    hip.set_option(k, v)
                tr = NativeTracer(sc, mode)
                for _ in range(2):
                    fb = tr().framebuffer(True)
                    assert np.abs(fb[..., :3] - ref[..., :3]).max() <= tol and np.array_equal(fb[..., 3], ref[..., 3]), opts
                    assert tr.stats["rays_closest"] == st.rays_closest and tr.stats["rays_any"] == st.rays_any, opts
                chains[tuple(opts.items())] = tr.stats["chains"]
                tr.close()
            finally:
                hip.set_option("defaults", 0)
        assert chains[(("hop_local", 2),)] <= chains[(("hop_local", 0),)], chains
        if sc.n_inst == 8:
            assert chains[(("hop_local", 2),)] < chains[(("hop_local", 0),)], chains  # the bunny grid: one chain instead of two
        tr = NativeTracer(sc, mode)  # default: timed
        seen = set()
        for _ in range(24):
            fb = tr().framebuffer(True)
            assert np.abs(fb[..., :3] - ref[..., :3]).max() <= tol and np.array_equal(fb[..., 3], ref[..., 3])
            assert tr.stats["rays_closest"] == st.rays_closest and tr.stats["rays_any"] == st.rays_any
            seen.add(tr.stats["chains"])
        tr.close()
        assert len(seen) >= 1
        for world in (2, 3):
            owner = [i % world for i in range(sc.n_inst)]
            refd, std = oracle_render_domain(sc, owner, world, 0 if mode == NORMALS_FLAT else 1)
            for hop in (0, 2, 3):
                for bsp in (False, True):
                    res = run_native_ranks(sc, owner, world, mode, bsp, opts=(("hop_local", hop),))
                    fb = res[0][0]
                    assert np.abs(fb[..., :3] - refd[..., :3]).max() <= tol and np.array_equal(fb[..., 3], refd[..., 3]), (world, hop, bsp)
                    assert sum(r[1]["rays_sent"] for r in res.values()) == std.rays_sent, (world, hop, bsp)
                    assert sum(r[1]["rays_closest"] for r in res.values()) == std.rays_closest and sum(r[1]["rays_any"] for r in res.values()) == std.rays_any, (world, hop, bsp)


def _random_scene(seed):
    """Two to seven instances of one to three random blobs, placed along and around the view axis so that their boxes overlap and rays
    cross several of them (hops, known misses, shadow rays between instances); one or two point lights; a small film."""
    from gravit_amd.layouts import default_material, point_light
    rng = np.random.default_rng(9000 + seed)
    meshes = []
    for _ in range(int(rng.integers(1, 4))):
        n_v, n_t = int(rng.integers(20, 300)), int(rng.integers(30, 900))
        v = (rng.normal(size=(n_v, 3)) * rng.uniform(0.1, 0.4, 3)).astype(np.float32)
        t = rng.integers(0, n_v, (n_t, 3)).astype(np.int32)
        t = t[(t[:, 0] != t[:, 1]) & (t[:, 1] != t[:, 2]) & (t[:, 0] != t[:, 2])]
        meshes.append(scenes.MeshData(v, np.ascontiguousarray(t), default_material(kd=rng.uniform(0.2, 0.9, 3))))
    n_inst = int(rng.integers(2, 8))
    inst_mesh = [int(rng.integers(0, len(meshes))) for _ in range(n_inst)]
    mats = [scenes.mat_translate_scale(rng.normal(size=3) * (0.5, 0.5, 0.9), rng.uniform(0.5, 1.6, 3)) for _ in range(n_inst)]
    lights = np.concatenate([point_light(rng.uniform(-3, 3, 3) + (0, 0, 3), rng.uniform(0.4, 1.0, 3)) for _ in range(int(rng.integers(1, 3)))])
    cam = scenes.Camera(eye=tuple(rng.normal(size=3) * 0.3 + (0.0, 0.0, 4.5)), focus=(0.0, 0.0, 0.0), up=(0.0, 1.0, 0.0), fov=float(np.radians(rng.uniform(25, 60))),
                        width=int(rng.integers(8, 24)) * 8 + int(rng.integers(0, 8)), height=int(rng.integers(60, 140)),
                        depth=2 if seed % 5 == 4 else 1)  # (every fifth scene with a bounce: several deposits per pixel)
    return scenes._assemble(meshes, inst_mesh, mats, lights, cam, "random %d" % seed)


@pytest.mark.parametrize("seed", range(10))
def test_random_scenes_through_the_native_schedulers(hip, seed):
    """Seeded fuzz of the scheduler loop: random overlapping instances (rays hop between them, meet known misses, their shadow rays cross
    other instances), on one rank against the oracle's restated Image scheduler and on two to four in-process ranks (asynchronous ticks
    and BSP rounds) against its restated DomainTracer: whole float framebuffers bit for bit (within 1e-5 where a bounce gives pixels several deposits), ray
    counts and rays sent equal."""
    sc = _random_scene(seed)
    mode = NORMALS_SMOOTH if seed % 2 else NORMALS_FLAT
    ref, st = oracle_render(sc, mode, nthreads=8)
    tr = NativeTracer(sc, mode)
    tol = 1e-5 if sc.camera.depth > 1 else 0.0  # bounces: a pixel gets several deposits, their float sum depends on arrival order
    for _ in range(9):  # (through the probing frames of finish_auto as well)
        fb = tr().framebuffer(True)
        assert np.abs(fb[..., :3] - ref[..., :3]).max() <= tol and np.array_equal(fb[..., 3], ref[..., 3])
        assert tr.stats["rays_closest"] == st.rays_closest and tr.stats["rays_any"] == st.rays_any
    tr.close()
    assert seed >= 10 or st.rays_closest > 500
    world = 2 + seed % 3
    owner = [i % world for i in range(sc.n_inst)]
    refd, std = oracle_render_domain(sc, owner, world, mode)
    assert seed >= 10 or std.rays_sent > 0  # (every one of the suite's ten scenes makes rays change rank: 42 .. 4,570 of them, in 3 .. 6 rounds)
    for bsp in (False, True):
        res = run_native_ranks(sc, owner, world, mode, bsp)
        assert np.abs(res[0][0][..., :3] - refd[..., :3]).max() <= tol
        assert sum(r[1]["rays_closest"] for r in res.values()) == std.rays_closest and sum(r[1]["rays_sent"] for r in res.values()) == std.rays_sent


@pytest.mark.parametrize("case", ["partly_off_film", "box_behind_the_eye_plane", "nothing_in_view", "jitter_window"])
def test_camera_rectangle_edge_cases(hip, case):
    """The camera filter enumerates only the film rectangle the (kept) instances' boxes project onto (sched.hip camera_keep_rect).
    Instances that leave the film on two sides, a box that reaches behind the eye plane (no bounded projection: the whole film is
    enumerated), a camera that sees nothing, sub-samples spread by a jitter window (whole film): images and ray counts are the
    oracle's, on one rank and on two."""
    sc = scenes.bunny_grid_scene(width=333, height=190)
    if case == "partly_off_film":
        sc.camera.eye, sc.camera.focus = (0.45, 0.2, 1.1), (0.45, 0.2, 0.0)
    elif case == "box_behind_the_eye_plane":
        sc.camera.eye, sc.camera.focus = (0.0, 0.1, 0.12), (0.3, 0.1, 0.0)  # among the bunnies' boxes, looking sideways
    elif case == "nothing_in_view":
        sc.camera.eye, sc.camera.focus = (0.0, 0.1, 1.6), (0.0, 0.1, 3.0)
    else:
        sc.camera.samples, sc.camera.jitter = 2, 1.0
    ref, st = oracle_render(sc, NORMALS_SMOOTH, nthreads=8)
    tr = NativeTracer(sc, NORMALS_SMOOTH)
    fb = tr().framebuffer(True)
    assert np.abs(fb[..., :3] - ref[..., :3]).max() <= (1e-5 if case == "jitter_window" else 0.0) and np.array_equal(fb[..., 3], ref[..., 3])
    assert tr.stats["rays_closest"] == st.rays_closest and tr.stats["rays_any"] == st.rays_any
    assert (st.rays_closest == 0) == (case == "nothing_in_view")
    tr.close()
    owner = [i % 2 for i in range(sc.n_inst)]
    res = run_native_ranks(sc, owner, 2, NORMALS_SMOOTH, False)
    refd, std = oracle_render_domain(sc, owner, 2, NORMALS_SMOOTH)
    assert np.abs(res[0][0][..., :3] - refd[..., :3]).max() <= (1e-5 if case == "jitter_window" else 0.0) and np.array_equal(res[0][0][..., 3], refd[..., 3])
    assert sum(r[1]["rays_closest"] for r in res.values()) == std.rays_closest and sum(r[1]["rays_sent"] for r in res.values()) == std.rays_sent


@pytest.mark.parametrize("launcher", ["torch.distributed.run", "none"])
def test_bench_script_native_multi_process_plumbing(hip, tmp_path, launcher):
    """bench.py's N > 1 branch with the NATIVE harness, two processes on this one GPU: RCCL refuses two ranks on one device, so
    --fake-comm leaves the communicator out (every rank renders every scene alone: the numbers mean nothing) -- what runs is the
    script's own plumbing around the library: rendezvous, the three scheduler variants, the config-4 and weak-soup legs with
    their gathers and reductions, the per-rank roofline, the JSON line.  (The exchange itself: the in-process ranks above.)"""
    import json
    import os
    import socket
    import subprocess
    import sys

    from tests.conftest import ROOT
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    # launcher "none": plain `python bench.py --gpus 2` with no RANK / WORLD_SIZE in the environment -- the script starts its two ranks itself (spawn_ranks: the
    # parent touches no GPU, the children are ordinary child processes) and hands rank 0's line through
    pre = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port)] if launcher != "none" else [sys.executable]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(pre + [os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-gpu", "--fake-comm", "--tris", "200000", "--weak-tris", "100000", "--width", "480", "--height", "270",
                              "--steps", "2", "--warmup", "1"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
    j = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["config"]["harness"] == "native" and j["roofline"]["frac"] > 0
    assert set(j["variants"]) == {"domain_async", "domain_bsp", "image_replicated", "domain_async_known_miss_shortcut"}
    c4, wk = j["config4_bunny_grid"], j["weak_soup"]
    assert c4["film"] == [1900, 1080] and c4["domain_async"]["rays_per_step"] > 100_000 and c4["domain_bsp"]["value"] > 0
    assert wk["tiles"] == 2 and wk["tris_per_tile"] == 100000 and wk["value"] > 0 and len(wk["roofline_per_rank"]) == 2 and all(r["frac"] > 0 for r in wk["roofline_per_rank"])
