"""one rank of tools/overlap_slabs.sh: the config-5 hall in 6 slabs on 3 processes (stand-in transport), payload_overlap_kb = argv[1]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch.distributed as dist
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
from gravit_amd import capi, scenes
from gravit_amd.layouts import NORMALS_FLAT
from gravit_amd.scheduler import Comm, NativeTracer
capi.init(0)
kb = int(sys.argv[1])
capi.set_option("payload_overlap_kb", kb)
uid = [Comm.unique_id() if rank == 0 else None]
dist.broadcast_object_list(uid, src=0)
comm = Comm.rccl(uid[0], rank, world)
one = scenes.cathedral_scene(512, 512, samples=2, depth=2, eye=(0.0, 1.5, 13.0), light=(0.0, 2.5, 12.0))
sc = scenes.split_into_domains(one, 6)
tr = NativeTracer(sc, NORMALS_FLAT, [i % world for i in range(sc.n_inst)], comm)
fb0 = None
for _ in range(2):
    tr()
dist.barrier()
t = time.perf_counter()
sums = {}
for _ in range(5):
    B = tr()
    for k, v in tr.stats.items():
        sums[k] = sums.get(k, 0) + v
dt = (time.perf_counter() - t) / 5
if rank == 0:
    fb = B.framebuffer(False)
    print("payload_overlap_kb = %d: %.2f ms per frame (stand-in transport: NOT a speed), %d ticks, %d rays sent per frame by rank 0 (%.1f KiB), %d inline; checksum of the composited frame %.6f, deposits %d"
          % (kb, dt * 1e3, sums["rounds"] // 5, sums["rays_sent"] // 5, sums["bytes_sent"] / 5 / 1024, sums["rays_inline"] // 5, float(fb[..., :3].astype(np.float64).sum()), int(fb[..., 3].sum())))
tr.close(); comm.close()
dist.barrier(); dist.destroy_process_group()
