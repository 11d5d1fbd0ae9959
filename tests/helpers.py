"""Shared helpers of the test-suite (checker side: may import the oracle)."""
import hashlib

import numpy as np

from oracle import orc


def oracle_meshes(scene):
    return [orc.Mesh(m.verts, m.tris, vnormals=m.vnormals, vcolors=m.vcolors, materials=m.materials, face_mat=m.face_mat,
                     mesh_mat=m.material) for m in scene.meshes]


def oracle_camera_rays(scene):
    c = scene.camera
    return orc.camera_rays(c.eye, c.focus, c.up, c.fov, c.width, c.height, c.samples, c.depth, c.jitter)


DEFAULT_RULE = "asis"  # tests/conftest.py switches the GPU tests to "strict"


def _by_rule(render, rule):
    """Which shuffleRays rule the checker restates: 'strict' -- the reference's hop-by-hop rule (TracerBase.h:392-400), the library's default and
    the GPU tests' default; 'shortcut' -- the library's opt-in known-miss shortcut (knob skip_known = 1), restated in the checker so that a
    device running it can be compared ray for ray; 'asis' -- the checker's flag left as the caller set it (the CPU tests, where the checker is
    also the backend under test)."""
    rule = rule or DEFAULT_RULE
    if rule == "asis":
        return render()
    assert rule in ("strict", "shortcut"), rule
    try:
        orc.set_skip_known_misses(rule == "shortcut")
        return render()
    finally:
        orc.set_skip_known_misses(False)


def oracle_render(scene, mode, nthreads=8, meshes=None, rule=None):
    meshes = meshes or oracle_meshes(scene)
    c = scene.camera
    rays = oracle_camera_rays(scene)
    return _by_rule(lambda: orc.render_image([meshes[i] for i in scene.inst_mesh], scene.m, scene.minv, scene.normi, scene.inst_lo, scene.inst_hi,
                                             scene.lights, rays, c.width, c.height, mode, nthreads), rule)


def oracle_render_domain(scene, owner, P, mode, nthreads=8, rule=None):
    meshes = oracle_meshes(scene)
    c = scene.camera
    rays = oracle_camera_rays(scene)
    return _by_rule(lambda: orc.render_domain([meshes[i] for i in scene.inst_mesh], scene.m, scene.minv, scene.normi, scene.inst_lo, scene.inst_hi,
                                              scene.lights, owner, P, rays, c.width, c.height, mode, nthreads), rule)


def sort_rays(r):
    """Canonical order of a ray list (the adapter's output order is unspecified)."""
    raw = np.ascontiguousarray(r).view(np.uint8).reshape(len(r), 80)[:, :68]
    keys = np.ascontiguousarray(raw).view(np.uint32).reshape(len(r), 17)
    idx = np.lexsort(keys.T[::-1])
    return r[idx]


def rays_equal_bits(a, b):
    if len(a) != len(b):
        return False
    ra = np.ascontiguousarray(a).view(np.uint8).reshape(len(a), 80)[:, :68]
    rb = np.ascontiguousarray(b).view(np.uint8).reshape(len(b), 80)[:, :68]
    return bool((ra == rb).all())


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def seeded_rays_at(mesh_lo, mesh_hi, n, seed, spread=0.3):
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(mesh_lo, np.float32), np.asarray(mesh_hi, np.float32)
    ext = hi - lo
    org = (lo - spread * ext + (1 + 2 * spread) * ext * rng.random((n, 3))).astype(np.float32)
    org[:, 2] = hi[2] + spread * ext[2] + 0.05
    tgt = (lo + ext * rng.random((n, 3))).astype(np.float32)
    d = tgt - org
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return org, d
