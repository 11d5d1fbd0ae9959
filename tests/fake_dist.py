"""In-process stand-in for torch.distributed used to run P DomainTracer ranks as P threads against ONE GPU
(the GPU box has a single device).  Implements exactly the calls DomainTracer makes: get_rank / get_world_size /
is_initialized, all_gather, P2POp + isend/irecv + batch_isend_irecv, reduce, ReduceOp.  TEST INFRASTRUCTURE."""
import threading


class _Req:
    def __init__(self, fn):
        self._fn = fn

    def wait(self):
        self._fn()


class ReduceOp:
    SUM = "sum"


class FakeWorld:
    def __init__(self, world):
        self.world = world
        self.barrier = threading.Barrier(world)
        self.slots = {}
        self.lock = threading.Lock()
        self.mail = {}
        self.cv = threading.Condition()

    def rank_view(self, rank):
        return FakeDist(self, rank)


class P2POp:
    def __init__(self, op, tensor, peer):
        self.op, self.tensor, self.peer = op, tensor, peer


class FakeDist:
    ReduceOp = ReduceOp
    P2POp = P2POp
    isend = "isend"
    irecv = "irecv"

    def __init__(self, world, rank):
        self.w, self.rank = world, rank

    def is_initialized(self):
        return True

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.w.world

    def all_gather(self, outs, t):
        with self.w.lock:
            self.w.slots[("ag", self.rank)] = t.clone()
        self.w.barrier.wait()
        for r in range(self.w.world):
            outs[r].copy_(self.w.slots[("ag", r)])
        self.w.barrier.wait()

    def batch_isend_irecv(self, ops):
        reqs = []
        for op in ops:
            if op.op == "isend":
                with self.w.cv:
                    self.w.mail.setdefault((self.rank, op.peer), []).append(op.tensor.clone())
                    self.w.cv.notify_all()
                reqs.append(_Req(lambda: None))
            else:
                def recv(op=op):
                    with self.w.cv:
                        while not self.w.mail.get((op.peer, self.rank)):
                            self.w.cv.wait(timeout=60)
                        src = self.w.mail[(op.peer, self.rank)].pop(0)
                    assert src.shape == op.tensor.shape, (src.shape, op.tensor.shape)
                    op.tensor.copy_(src)
                reqs.append(_Req(recv))
        return reqs

    def reduce(self, t, dst=0, op=None):
        with self.w.lock:
            self.w.slots[("red", self.rank)] = t.clone()
        self.w.barrier.wait()
        if self.rank == dst:
            acc = self.w.slots[("red", 0)].clone()
            for r in range(1, self.w.world):
                acc += self.w.slots[("red", r)]
            t.copy_(acc)
        self.w.barrier.wait()
