"""Checker backend for gravit_amd.scheduler: the same backend interface as HipBackend, implemented with the
CPU oracle and numpy.  TEST INFRASTRUCTURE: lets the multi-rank control flow of DomainTracer (exchange,
termination, composite) run under gloo on CPU.  Never used by the product."""
import numpy as np

from oracle import orc


def cat_rays(parts):
    """np.concatenate would re-pack the 80-byte record to 64 bytes; keep the reference layout."""
    n = sum(len(p) for p in parts)
    out = np.zeros(n, orc.RAY_DTYPE)
    o = 0
    for p in parts:
        out[o:o + len(p)] = p
        o += len(p)
    return out


class OracleBackend:
    def __init__(self, scene, normal_mode=0, owned=None):
        self.scene = scene
        self.normal_mode = normal_mode
        self.n_inst = scene.n_inst
        self.owned = [True] * self.n_inst if owned is None else list(owned)
        self.order = orc.toplevel_order(scene.inst_lo, scene.inst_hi)
        self.meshes = {}
        for i in range(self.n_inst):
            mi = scene.inst_mesh[i]
            if self.owned[i] and mi not in self.meshes:
                m = scene.meshes[mi]
                self.meshes[mi] = orc.Mesh(m.verts, m.tris, mesh_mat=m.material)
        self.queues = [np.zeros(0, orc.RAY_DTYPE) for _ in range(self.n_inst)]
        cam = scene.camera
        self.fb = np.zeros((cam.height * cam.width, 4), np.float32)
        self.calls = 0
        self.rays_closest = self.rays_any = 0
        self._fbt = None

    def begin_frame(self):
        self.fb[:] = 0
        self.queues = [np.zeros(0, orc.RAY_DTYPE) for _ in range(self.n_inst)]
        self.calls = 0

    def _shuffle(self, rays, frm, keep_mask=None):
        if len(rays) == 0:
            return
        rays = np.ascontiguousarray(rays).copy()
        nxt = orc.shuffle_step(self.scene.inst_lo, self.scene.inst_hi, self.order, rays, frm)  # origins advanced in place (TracerBase.h:393)
        hit = nxt >= 0
        moved = rays[hit]
        for q in np.unique(nxt[hit]):
            if keep_mask is None or keep_mask[q]:
                self.queues[q] = cat_rays([self.queues[q], moved[nxt[hit] == q]])
        dep = rays[(~hit) & (rays["type"] == 1)]
        dep = dep[np.sqrt((dep["color"] * dep["color"]).sum(1)) > 0]
        if len(dep):
            np.add.at(self.fb[:, :3], dep["id"], dep["color"] * dep["w"][:, None])
            np.add.at(self.fb[:, 3], dep["id"], 1.0)

    def generate_and_filter(self, keep_mask=None):
        c = self.scene.camera
        rays = orc.camera_rays(c.eye, c.focus, c.up, c.fov, c.width, c.height, c.samples, c.depth, c.jitter)
        self._shuffle(rays, -1, keep_mask)

    def queue_sizes(self):
        return [len(q) for q in self.queues]

    def trace_and_shuffle(self, inst):
        s = self.scene
        rays = np.ascontiguousarray(self.queues[inst])
        moved = self.meshes[s.inst_mesh[inst]].trace(rays, s.m[inst], s.minv[inst], s.normi[inst], s.lights, self.normal_mode, self.calls, 2)
        c, a = orc.trace_counts()
        self.rays_closest += c
        self.rays_any += a
        self.calls += 1
        self.queues[inst] = np.zeros(0, orc.RAY_DTYPE)
        self._shuffle(moved, inst, None)

    def export_wire(self, insts, torch, device):
        parts = [self.queues[i] for i in insts]
        for i in insts:
            self.queues[i] = np.zeros(0, orc.RAY_DTYPE)
        raw = cat_rays(parts)
        return torch.from_numpy(np.ascontiguousarray(raw).view(np.float32).reshape(-1, 20).copy())

    def append_wire(self, inst, buf, off, n):
        if n:
            rays = buf[off:off + n].contiguous().numpy().view(orc.RAY_DTYPE).reshape(-1)
            self.queues[inst] = cat_rays([self.queues[inst], rays])

    def fb_tensor(self, torch, device):
        self._fbt = torch.from_numpy(self.fb.reshape(-1))
        return self._fbt

    def framebuffer(self, clamp=True):
        c = self.scene.camera
        fb = self.fb.reshape(c.height, c.width, 4).copy()
        if clamp:
            fb[..., :3] = np.minimum(fb[..., :3], 1.0)
        return fb

    def sync(self):
        pass
