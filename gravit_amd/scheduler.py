"""The two live GraviT schedulers around Adapter::trace, re-hosted on device-resident ray queues.

  Tracer<ImageScheduler>::operator()   src/gvt/render/algorithm/ImageTracer.h:127-269
  Tracer<DomainScheduler>::operator()  src/gvt/render/algorithm/DomainTracer.h:115-496
  AbstractTrace::shuffleRays           src/gvt/render/algorithm/TracerBase.h:325-414

The loops are the reference's (pick the fullest queue, trace it, shuffle the moved rays; Domain: trace
until the local queues are dry, exchange, test for termination).  What changes is where the data
lives: queues, the top-level test, the framebuffer and the adapter all stay in HBM; under the Domain
scheduler the MPI ray exchange (DomainTracer.h:370-496) becomes point-to-point RCCL over xGMI through
torch.distributed, carrying the reference's own 80-byte Ray image (actor/Ray.h:161-174).

The device work goes through a small backend object so that the multi-rank control flow can be
exercised on CPU (gloo) in the tests with a checker backend; the product backend is HipBackend.
"""
import numpy as np

from . import capi
from .adapter import FrameBuffer, HipMeshAdapter, RayQueue, TopLevel, camera_generate
from .layouts import LIGHT_DTYPE, NORMALS_FLAT


class HipBackend:
    """One rank's device state: adapters (cached per mesh like adapterCache, ImageTracer.h:184-233),
    per-instance queues, top-level set, framebuffer."""

    def __init__(self, scene, normal_mode=NORMALS_FLAT, owned=None):
        self.scene = scene
        self.normal_mode = normal_mode
        self.n_inst = scene.n_inst
        self.owned = [True] * self.n_inst if owned is None else list(owned)
        self.top = TopLevel(scene.inst_lo, scene.inst_hi)
        self.queues = [RayQueue() for _ in range(self.n_inst)]
        self.q_cam = RayQueue()
        self.q_moved = RayQueue()
        cam = scene.camera
        self.fb = FrameBuffer(cam.width, cam.height)
        self.adapter_cache = {}
        self.calls = 0
        for i in range(self.n_inst):  # adapters are built lazily in the reference; here up front, off the frame clock
            if self.owned[i]:
                self.adapter(i)

    def adapter(self, inst):
        mi = self.scene.inst_mesh[inst]
        if mi not in self.adapter_cache:
            self.adapter_cache[mi] = HipMeshAdapter(self.scene.meshes[mi], self.normal_mode)
        return self.adapter_cache[mi]

    # ---- frame steps
    def begin_frame(self):
        self.fb.clear()
        for q in self.queues:
            q.clear()
        self.calls = 0

    def generate_and_filter(self, keep_mask=None):
        """camera rays -> FilterRaysLocally: shuffleRays(rays,-1) (ImageTracer.h:111-125) or, with a keep mask,
        shuffleDropRays (DomainTracer.h:148-183)."""
        import ctypes as C

        cam = self.scene.camera
        pod = capi.CameraPod((C.c_float * 3)(*cam.eye), (C.c_float * 3)(*cam.focus), (C.c_float * 3)(*cam.up), cam.fov, cam.width, cam.height,
                             cam.samples, cam.depth, cam.jitter)
        queues = (C.c_void_p * max(1, self.n_inst))(*[q.h for q in self.queues])
        km = None if keep_mask is None else np.ascontiguousarray(keep_mask, dtype=np.uint8)
        capi.check(capi.load().gvt_hip_camera_filter(self.top.h, C.byref(pod), C.c_int(8), queues, capi.ptr(km)), "gvt_hip_camera_filter")

    def queue_sizes(self):
        return [len(q) for q in self.queues]

    def image_frame(self):
        """Tracer<ImageScheduler>::operator() run natively inside the library (gvt_hip_image_frame); returns the number of adapter calls."""
        import ctypes as C

        s = self.scene
        cam = s.camera
        pod = capi.CameraPod((C.c_float * 3)(*cam.eye), (C.c_float * 3)(*cam.focus), (C.c_float * 3)(*cam.up), cam.fov, cam.width, cam.height,
                             cam.samples, cam.depth, cam.jitter)
        meshes = (C.c_void_p * max(1, self.n_inst))(*[self.adapter(i).h for i in range(self.n_inst)])
        queues = (C.c_void_p * max(1, self.n_inst))(*[q.h for q in self.queues])
        lights = np.ascontiguousarray(s.lights, dtype=LIGHT_DTYPE)  # (a concatenation of records drops the dtype's padding: always the ABI's 64-byte layout)
        calls = C.c_uint64(0)
        capi.check(capi.load().gvt_hip_image_frame(
            self.top.h, meshes, capi.ptr(capi.f32(s.m)), capi.ptr(capi.f32(s.minv)), capi.ptr(capi.f32(s.normi)), C.c_size_t(self.n_inst),
            capi.ptr(lights), C.c_size_t(len(lights)), C.c_int(self.normal_mode), C.byref(pod), queues, self.q_cam.h, self.q_moved.h, self.fb.h,
            C.byref(calls)), "gvt_hip_image_frame")
        return calls.value

    def trace_and_shuffle(self, inst):
        s = self.scene
        self.adapter(inst).trace_queue(self.queues[inst], self.q_moved, s.m[inst], s.minv[inst], s.normi[inst], s.lights, seed=self.calls,
                                       sink=(self.top, inst, self.fb))
        self.calls += 1
        self.top.shuffle(self.q_moved, inst, self.queues, self.fb, None)

    # ---- exchange (wire format: the reference's 80-byte Ray image)
    def export_wire(self, insts, torch, device):
        """Concatenate the queues `insts` into one wire tensor [n, 20] f32 and clear them."""
        sizes = [len(self.queues[i]) for i in insts]
        total = sum(sizes)
        buf = torch.empty((total, 20), dtype=torch.float32, device=device)
        off = 0
        for i, n in zip(insts, sizes):
            if n:
                self.queues[i].export_device(buf.data_ptr() + off * 80, n)
                self.queues[i].clear()
            off += n
        return buf

    def append_wire(self, inst, buf, off, n):
        if n:
            self.queues[inst].append_device(buf.data_ptr() + off * 80, n)

    def fb_tensor(self, torch, device):
        """torch view of the framebuffer's device memory (un-clamped sums) for the composite reduce."""
        cam = self.scene.camera

        class _View:
            __cuda_array_interface__ = {"shape": (cam.height * cam.width * 4,), "typestr": "<f4",
                                        "data": (self.fb.device_ptr(), False), "version": 2}

        return torch.as_tensor(_View(), device=device)

    def framebuffer(self, clamp=True):
        return self.fb.download(clamp)

    def sync(self):
        capi.synchronize()


class Context:
    """gvt_hip_ctx: a stream + scratch + counters of its own, current for the calling thread (several ranks in one process).
    `with Context(0): ...` in the thread that uses it; close() releases its stream, spill arena and scratch."""

    def __init__(self, device=0):
        import ctypes as C

        self.lib = capi.load()
        self.h = C.c_void_p(self.lib.gvt_hip_ctx_create(C.c_int(device)))
        if not self.h:
            raise capi.GvtHipError("gvt_hip_ctx_create: " + capi.last_error())
        capi.check(self.lib.gvt_hip_ctx_make_current(self.h), "gvt_hip_ctx_make_current")

    def close(self):
        if getattr(self, "h", None):
            self.lib.gvt_hip_ctx_make_current(None)
            self.lib.gvt_hip_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        import gc

        gc.collect()  # wrappers created under this context release their device memory before its stream goes
        self.close()
        return False


class Comm:
    """gvt_hip_comm: one rank's endpoint of the ray exchange.  Comm.rccl(dist-like broadcast) for one process per GPU,
    Comm.local(hub, rank) for in-process ranks."""

    def __init__(self, h):
        self.lib = capi.load()
        self.h = h
        self.rank = self.lib.gvt_hip_comm_rank(h)
        self.world = self.lib.gvt_hip_comm_world(h)

    @staticmethod
    def unique_id():
        import ctypes as C

        buf = (C.c_ubyte * 128)()
        capi.check(capi.load().gvt_hip_comm_unique_id(buf), "gvt_hip_comm_unique_id")
        return bytes(buf)

    @classmethod
    def rccl(cls, uid, rank, world):
        import ctypes as C

        buf = (C.c_ubyte * 128).from_buffer_copy(uid)
        h = C.c_void_p(capi.load().gvt_hip_comm_create(buf, C.c_int(rank), C.c_int(world)))
        if not h:
            raise capi.GvtHipError("gvt_hip_comm_create: " + capi.last_error())
        return cls(h)

    @classmethod
    def local(cls, hub, rank):
        import ctypes as C

        h = C.c_void_p(capi.load().gvt_hip_comm_create_local(C.c_void_p(hub), C.c_int(rank)))
        if not h:
            raise capi.GvtHipError("gvt_hip_comm_create_local: " + capi.last_error())
        return cls(h)

    @property
    def count(self):
        """ranks the transport itself reports (ncclCommCount; the hub's world)"""
        return self.lib.gvt_hip_comm_count(self.h)

    @property
    def reserved_cus(self):
        """compute units reserved for the communicator's own stream (knob comm_cus when it was created)"""
        return self.lib.gvt_hip_comm_reserved_cus(self.h)

    def set_deadline_ms(self, ms):
        import ctypes as C

        capi.check(self.lib.gvt_hip_comm_set_deadline_ms(self.h, C.c_int(int(ms))), "gvt_hip_comm_set_deadline_ms")

    def selftest(self, nbytes=1 << 20):
        import ctypes as C

        capi.check(self.lib.gvt_hip_comm_selftest(self.h, C.c_size_t(nbytes)), "gvt_hip_comm_selftest")

    def close(self):
        if getattr(self, "h", None):
            self.lib.gvt_hip_comm_destroy(self.h)
            self.h = None


class NativeTracer:
    """gvt_hip_tracer: Tracer<ImageScheduler> / Tracer<DomainScheduler> as a native loop inside libgvt_hip.so (csrc/domain.hip): rounds
    of one merged launch chain over all local queues, one host synchronisation per round, ray exchange over RCCL (or the in-process
    transport).  owner[i] = rank of instance i (mpiInstanceMap); comm=None: one rank."""

    def __init__(self, scene, normal_mode=NORMALS_FLAT, owner=None, comm=None, backend=None, replicate=False):
        import ctypes as C

        self.lib = capi.load()
        self.scene = scene
        self.comm = comm
        rank = comm.rank if comm is not None else 0
        self.owner = [0] * scene.n_inst if owner is None else list(owner)
        owned = [True] * scene.n_inst if replicate else [o == rank for o in self.owner]  # replicate: every mesh on every rank (Image scheduler)
        self.backend = backend or HipBackend(scene, normal_mode, owned)
        B = self.backend
        cam = scene.camera
        pod = capi.CameraPod((C.c_float * 3)(*cam.eye), (C.c_float * 3)(*cam.focus), (C.c_float * 3)(*cam.up), cam.fov, cam.width, cam.height,
                             cam.samples, cam.depth, cam.jitter)
        meshes = (C.c_void_p * max(1, scene.n_inst))(*[B.adapter(i).h if owned[i] else None for i in range(scene.n_inst)])
        lights = np.ascontiguousarray(scene.lights, dtype=LIGHT_DTYPE)  # (np.concatenate of light records packs them to 48 bytes: always the ABI's 64-byte layout)
        self.h = C.c_void_p(self.lib.gvt_hip_tracer_create(
            B.top.h, meshes, capi.ptr(capi.f32(scene.m)), capi.ptr(capi.f32(scene.minv)), capi.ptr(capi.f32(scene.normi)), C.c_size_t(scene.n_inst),
            capi.ptr(lights), C.c_size_t(len(lights)), C.c_int(normal_mode), C.byref(pod), B.fb.h))
        if not self.h:
            raise capi.GvtHipError("gvt_hip_tracer_create: " + capi.last_error())
        own = np.ascontiguousarray(self.owner, dtype=np.int32)
        capi.check(self.lib.gvt_hip_tracer_set_domains(self.h, capi.ptr(own), comm.h if comm is not None else None), "gvt_hip_tracer_set_domains")
        self.frame_stats = None  # the last frame's gvt_hip_frame_stats as the library filled it (`stats`: the same as a dict, made on access)

    @property
    def stats(self):
        return self.frame_stats.as_dict() if self.frame_stats is not None else {}

    def __call__(self, bsp=False, composite=True, full_reduce=False, image=False):
        import ctypes as C

        st = capi.FrameStats()
        flags = ((capi.FRAME_BSP if bsp else 0) | (0 if composite else capi.FRAME_NO_COMPOSITE) | (capi.FRAME_FULL_REDUCE if full_reduce else 0) |
                 (capi.FRAME_IMAGE if image else 0))
        capi.check(self.lib.gvt_hip_tracer_frame(self.h, C.c_int(flags), C.byref(st)), "gvt_hip_tracer_frame")
        self.frame_stats = st
        return self.backend

    render = __call__

    def set_camera(self, cam):
        """gvt_hip_tracer_set_camera: the next frame's camera (a moving camera re-uses the tracer, its queues and its framebuffer: same film size)."""
        import ctypes as C

        if (cam.width, cam.height) != (self.scene.camera.width, self.scene.camera.height):
            raise ValueError("set_camera: the film size belongs to the tracer's framebuffer")
        pod = capi.CameraPod((C.c_float * 3)(*cam.eye), (C.c_float * 3)(*cam.focus), (C.c_float * 3)(*cam.up), cam.fov, cam.width, cam.height,
                             cam.samples, cam.depth, cam.jitter)
        capi.check(self.lib.gvt_hip_tracer_set_camera(self.h, C.byref(pod)), "gvt_hip_tracer_set_camera")

    def close(self):
        if getattr(self, "h", None):
            self.lib.gvt_hip_tracer_destroy(self.h)
            self.h = None

    __del__ = close


def _pick_fullest(sizes, allowed=None):
    """ImageTracer.h:159-173 / DomainTracer.h:235-241: first queue with the strictly largest size."""
    target, cnt = -1, 0
    for i, n in enumerate(sizes):
        if (allowed is None or allowed[i]) and n > cnt:
            cnt, target = n, i
    return target


class ImageTracer:
    """Tracer<ImageScheduler>, one rank (ImageTracer.h:127-269)."""

    def __init__(self, scene, normal_mode=NORMALS_FLAT, backend=None, native=True):
        self.backend = backend or HipBackend(scene, normal_mode)
        self.adapter_calls = 0
        self.native = native  # run the loop natively (gvt_hip_image_frame) when the backend offers it

    def __call__(self):
        B = self.backend
        if hasattr(B, "image_frame") and self.native:
            self.adapter_calls = B.image_frame()  # the same loop, inside libgvt_hip.so
            return B
        B.begin_frame()  # clearBuffer
        B.generate_and_filter(None)  # FilterRaysLocally
        self.adapter_calls = 0
        while True:
            target = _pick_fullest(B.queue_sizes())
            if target < 0:
                break
            B.trace_and_shuffle(target)  # adapter->trace + shuffleRays(moved_rays, instTarget)
            self.adapter_calls += 1
        return B

    render = __call__


class DomainTracer:
    """Tracer<DomainScheduler> over torch.distributed (DomainTracer.h:185-496).

    owner[i] = rank that holds instance i's data (mpiInstanceMap, DomainTracer.h:115-144).  Per round:
    trace until the local queues are dry, then SendRays: one all-gather of the per-queue outgoing counts
    (replaces the count exchange :409-415 and, because every rank then knows how many rays are in
    flight, also the gather/scatter termination test :337-349) and one point-to-point payload exchange
    (:433-463).  The frame ends with a sum-reduce of the float framebuffers to rank 0
    (IceTComposite::composite, IceTComposite.cpp:84-101).
    """

    def __init__(self, scene, owner, dist, torch, comm_device, normal_mode=NORMALS_FLAT, backend=None, overlap=False):
        self.overlap = overlap  # post each exchange before the next local adapter call and complete it afterwards
        self.scene = scene
        self.owner = list(owner)
        self.dist, self.torch, self.dev = dist, torch, comm_device
        self.on_cuda = torch.device(comm_device).type == "cuda"
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.owned = [o == self.rank for o in self.owner]
        self.backend = backend or HipBackend(scene, normal_mode, self.owned)
        self.rounds = 0
        self.rays_sent = 0
        self.adapter_calls = 0

    def _post_exchange(self, counts, sizes):
        """SendRays payload (:433-463) as non-blocking point-to-point operations; returns (requests, receive buffers, send buffers)."""
        B, torch, dist = self.backend, self.torch, self.dist
        n_inst = len(self.owner)
        ops, send_bufs, recv_bufs = [], {}, {}
        for p in range(self.world):
            if p == self.rank:
                continue
            outq = [i for i in range(n_inst) if self.owner[i] == p and sizes[i] > 0]
            if outq:
                send_bufs[p] = B.export_wire(outq, torch, self.dev)
                self.rays_sent += int(send_bufs[p].shape[0])
                ops.append(dist.P2POp(dist.isend, send_bufs[p], p))
            n_in = int(sum(counts[p][i] for i in range(n_inst) if self.owned[i]))
            if n_in:
                recv_bufs[p] = torch.empty((n_in, 20), dtype=torch.float32, device=self.dev)
                ops.append(dist.P2POp(dist.irecv, recv_bufs[p], p))
        B.sync()  # wire buffers were filled on the adapter stream
        reqs = dist.batch_isend_irecv(ops) if ops else []
        return reqs, recv_bufs, send_bufs

    def _complete_exchange(self, posted, counts):
        reqs, recv_bufs, _send_bufs = posted
        B = self.backend
        n_inst = len(self.owner)
        for req in reqs:
            req.wait()
        if self.on_cuda:
            self.torch.cuda.current_stream().synchronize()
        for p, buf in recv_bufs.items():  # unpack into queue[q] (:466-481)
            off = 0
            for i in range(n_inst):
                if self.owned[i] and counts[p][i]:
                    B.append_wire(i, buf, off, int(counts[p][i]))
                    off += int(counts[p][i])

    def _frame_overlapped(self):
        """The same frame with the ray exchange overlapped with traversal: every iteration all ranks exchange the outgoing counts (and
        whether they still hold local work), post the payload transfers of what has accumulated so far, run ONE local adapter call
        while those move (RCCL on its own stream), then complete the transfers.  Ends when no rank has local work and nothing is in
        flight -- the vote of the reference's asynchronous tracer (tracer/Domain/DomainTracer.cpp:109-192) folded into the count
        exchange.  Same rays, same image; only the interleaving differs."""
        B, torch, dist = self.backend, self.torch, self.dist
        n_inst = len(self.owner)
        B.begin_frame()
        B.generate_and_filter(np.array(self.owned, np.uint8))  # shuffleDropRays
        self.rounds = self.rays_sent = self.adapter_calls = 0
        while True:
            sizes = B.queue_sizes()
            target = _pick_fullest(sizes, self.owned)
            if self.world == 1:
                if target < 0:
                    break
                B.trace_and_shuffle(target)
                self.adapter_calls += 1
                continue
            mine = torch.tensor([0 if self.owned[i] else sizes[i] for i in range(n_inst)] + [1 if target >= 0 else 0], dtype=torch.int64,
                                device=self.dev)
            allc = torch.empty((self.world, n_inst + 1), dtype=torch.int64, device=self.dev)
            rows = [allc[r] for r in range(self.world)]
            dist.all_gather(rows, mine)
            counts = torch.stack(rows).cpu().numpy()
            if int(counts[:, :n_inst].sum()) == 0 and int(counts[:, n_inst].sum()) == 0:
                break
            posted = self._post_exchange(counts, sizes)
            if target >= 0:  # one local adapter call while the payload moves
                B.trace_and_shuffle(target)
                self.adapter_calls += 1
            self._complete_exchange(posted, counts)
            self.rounds += 1
        return B

    def __call__(self):
        if self.overlap:
            return self._frame_overlapped()
        B, torch, dist = self.backend, self.torch, self.dist
        n_inst = len(self.owner)
        B.begin_frame()
        B.generate_and_filter(np.array(self.owned, np.uint8))  # shuffleDropRays
        self.rounds = self.rays_sent = self.adapter_calls = 0
        while True:
            while True:  # local work until dry (:228-326)
                target = _pick_fullest(B.queue_sizes(), self.owned)
                if target < 0:
                    break
                B.trace_and_shuffle(target)
                self.adapter_calls += 1
            self.rounds += 1
            if self.world == 1:
                break
            # ---- SendRays (:370-496)
            sizes = B.queue_sizes()
            mine = torch.tensor([0 if self.owned[i] else sizes[i] for i in range(n_inst)], dtype=torch.int64, device=self.dev)
            allc = torch.empty((self.world, n_inst), dtype=torch.int64, device=self.dev)
            rows = [allc[r] for r in range(self.world)]
            dist.all_gather(rows, mine)
            counts = torch.stack(rows).cpu().numpy()
            in_flight = int(counts.sum())
            if in_flight == 0:  # all_done (:337-349)
                break
            ops, send_bufs, recv_bufs = [], {}, {}
            for p in range(self.world):
                if p == self.rank:
                    continue
                outq = [i for i in range(n_inst) if self.owner[i] == p and sizes[i] > 0]
                if outq:
                    send_bufs[p] = B.export_wire(outq, torch, self.dev)
                    self.rays_sent += int(send_bufs[p].shape[0])
                    ops.append(dist.P2POp(dist.isend, send_bufs[p], p))
                n_in = int(sum(counts[p][i] for i in range(n_inst) if self.owned[i]))
                if n_in:
                    recv_bufs[p] = torch.empty((n_in, 20), dtype=torch.float32, device=self.dev)
                    ops.append(dist.P2POp(dist.irecv, recv_bufs[p], p))
            B.sync()  # wire buffers were filled on the adapter stream
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            if self.on_cuda:
                torch.cuda.current_stream().synchronize()
            for p, buf in recv_bufs.items():  # unpack into queue[q] (:466-481)
                off = 0
                for i in range(n_inst):
                    if self.owned[i] and counts[p][i]:
                        B.append_wire(i, buf, off, int(counts[p][i]))
                        off += int(counts[p][i])
        return B

    render = __call__

    def composite(self, download=True, rows_only=True):
        """Sum of the per-rank float framebuffers on rank 0 (in place, in HBM), then -- if download -- the localAdd clamp and the
        copy to the host.  Returns (H,W,4) on rank 0, None elsewhere or when download is False.

        rows_only: a rank's deposits lie where its domains project, so instead of a sum-reduce of whole frames (IceT's job in the
        reference, IceTComposite.cpp:84-101) every rank sends rank 0 just the bounding rectangle of the pixels it wrote to: one
        all-gather of the rectangles, point-to-point transfers, an add per rectangle.  Same sums, rank order instead of tree order."""
        B, torch, dist = self.backend, self.torch, self.dist
        cam = self.scene.camera
        if self.world == 1:
            return B.framebuffer(True) if download else None
        B.sync()
        t = B.fb_tensor(torch, self.dev)
        if not rows_only:
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
        else:
            img = t.view(cam.height, cam.width, 4)
            lit = img[..., 3] > 0  # every deposit counts in the fourth component (localAdd adds 1.0)
            ys = torch.nonzero(lit.any(dim=1)).flatten()
            xs = torch.nonzero(lit.any(dim=0)).flatten()
            box = [int(ys[0].item()), int(ys[-1].item()) + 1, int(xs[0].item()), int(xs[-1].item()) + 1] if ys.numel() else [0, 0, 0, 0]
            mine = torch.tensor(box, dtype=torch.int64, device=self.dev)
            allr = torch.empty((self.world, 4), dtype=torch.int64, device=self.dev)
            parts = [allr[r] for r in range(self.world)]
            dist.all_gather(parts, mine)
            boxes = torch.stack(parts).cpu().numpy()
            ops, bufs = [], {}
            if self.rank == 0:
                for p in range(1, self.world):
                    y0, y1, x0, x1 = (int(v) for v in boxes[p])
                    if y1 > y0:
                        bufs[p] = torch.empty((y1 - y0, x1 - x0, 4), dtype=torch.float32, device=self.dev)
                        ops.append(dist.P2POp(dist.irecv, bufs[p], p))
            elif box[1] > box[0]:
                bufs[0] = img[box[0]:box[1], box[2]:box[3]].contiguous()
                ops.append(dist.P2POp(dist.isend, bufs[0], 0))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            if self.rank == 0:
                for p, buf in bufs.items():
                    y0, y1, x0, x1 = (int(v) for v in boxes[p])
                    img[y0:y1, x0:x1] += buf
        if self.on_cuda:
            torch.cuda.current_stream().synchronize()
        if self.rank != 0 or not download:
            return None
        return B.framebuffer(True).reshape(cam.height, cam.width, 4)
