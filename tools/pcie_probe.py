"""PCIe and host-copy rates on the GPU box, for sizing the host-RayVector path (gvt_hip_trace): pinned vs pageable copies of a
rayList-sized buffer in each direction, both directions at once, and CPU memcpy into / out of pinned memory with 1..8 threads."""
import threading, time
import numpy as np
import torch
N = 83 * 1024 * 1024
dev = torch.device("cuda", 0)
d = torch.empty(N, dtype=torch.uint8, device=dev); d2 = torch.empty(N, dtype=torch.uint8, device=dev)
pin_a = torch.empty(N, dtype=torch.uint8).pin_memory(); pin_b = torch.empty(N, dtype=torch.uint8).pin_memory()
pg_a = torch.empty(N, dtype=torch.uint8); pg_b = torch.empty(N, dtype=torch.uint8); pg_a.fill_(1); pg_b.fill_(2)
def t(fn, reps=5):
    fn(); torch.cuda.synchronize(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best
gb = N / 1e9
print("pinned   H2D %.1f GB/s   D2H %.1f GB/s" % (gb / t(lambda: d.copy_(pin_a, non_blocking=True)), gb / t(lambda: pin_b.copy_(d, non_blocking=True))))
print("pageable H2D %.1f GB/s   D2H %.1f GB/s" % (gb / t(lambda: d.copy_(pg_a)), gb / t(lambda: pg_b.copy_(d))))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(s1): d.copy_(pin_a, non_blocking=True)
    with torch.cuda.stream(s2): pin_b.copy_(d2, non_blocking=True)
print("pinned both directions at once: %.1f GB/s each" % (gb / t(both)))
a, b = pin_a.numpy(), pg_b.numpy()
for nth in (1, 2, 4, 8):
    cuts = np.linspace(0, N, nth + 1).astype(np.int64)
    def run(src, dst):
        th = [threading.Thread(target=lambda i=i: np.copyto(dst[cuts[i]:cuts[i + 1]], src[cuts[i]:cuts[i + 1]])) for i in range(nth)]
        [x.start() for x in th]; [x.join() for x in th]
    t0 = time.perf_counter(); run(a, b); t1 = time.perf_counter(); run(b, a); t2 = time.perf_counter()
    run(a, b); t3 = time.perf_counter(); run(b, a); t4 = time.perf_counter()
    print("memcpy %d threads: pinned->pageable %.1f GB/s, pageable->pinned %.1f GB/s" % (nth, gb / (t3 - t2), gb / (t4 - t3)))
