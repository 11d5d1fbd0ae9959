"""The GraviT-side binding itself: oracle/_ref/dropin_demo is gravit_amd/host/HipMeshAdapter.cpp compiled against the
reference's OWN headers and sources (Adapter.h, Ray, Mesh, PointLight, glm) and linked with libgvt_hip.so.  It calls
HipMeshAdapter::trace through a gvt::render::Adapter* with a std::vector<gvt::render::actor::Ray>; the result must be
the oracle's.  The binary is built in the build container (make -C oracle dropin) and travels with the tree."""
import os
import subprocess

import numpy as np
import pytest

from gravit_amd import scenes
from oracle import orc
from tests.conftest import GOLDEN, ROOT
from tests.helpers import oracle_camera_rays, rays_equal_bits, sort_rays

pytestmark = pytest.mark.gpu
DEMO = os.path.join(ROOT, "oracle", "_ref", "dropin_demo")


@pytest.mark.skipif(not os.path.exists(DEMO), reason="oracle/_ref/dropin_demo not built (needs the GraviT tree at build time)")
@pytest.mark.parametrize("mode", [0, 1])
def test_reference_types_through_the_virtual_interface(tmp_path, mode, hip):
    out, rin = tmp_path / "out.bin", tmp_path / "rays.bin"
    W, H = 160, 120
    sc = scenes.bunny_scene(W, H)
    rays = oracle_camera_rays(sc)
    rays.tofile(rin)
    r = subprocess.run([DEMO, os.path.join(GOLDEN, "bunny.obj"), str(rin), str(mode), str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    raw = np.fromfile(out, np.uint8)
    n = int(raw[:8].view(np.uint64)[0])
    moved = raw[8:8 + 80 * n].view(orc.RAY_DTYPE)
    ray_list = raw[8 + 80 * n:].view(orc.RAY_DTYPE)
    assert len(ray_list) == W * H
    om = orc.Mesh(sc.meshes[0].verts, sc.meshes[0].tris, mesh_mat=sc.meshes[0].material)
    exp = om.trace(rays, sc.m[0], sc.minv[0], sc.normi[0], sc.lights, mode)
    assert n == len(exp) and n > 1000
    assert rays_equal_bits(sort_rays(moved.copy()), sort_rays(exp))
    assert rays_equal_bits(ray_list.copy(), rays)


@pytest.mark.skipif(not os.path.exists(DEMO), reason="oracle/_ref/dropin_demo not built (needs the GraviT tree at build time)")
def test_dropin_path_at_1080p_with_its_rate(tmp_path, hip):
    """The literal drop-in path at the benchmark's film size: 2,073,600 gvt::render::actor::Ray in a std::vector through
    gvt::render::Adapter::trace (virtual call, HipMeshAdapter.cpp, gvt_hip_trace: H2D, trace, D2H of rayList and moved rays).
    Result bit-identical to the oracle; the binary reports the call's wall time, which DESIGN.md quotes."""
    out, rin = tmp_path / "out.bin", tmp_path / "rays.bin"
    W, H = 1920, 1080
    sc = scenes.bunny_scene(W, H)
    rays = oracle_camera_rays(sc)
    rays.tofile(rin)
    r = subprocess.run([DEMO, os.path.join(GOLDEN, "bunny.obj"), str(rin), "0", str(out), "3"], capture_output=True, text=True, timeout=600)  # 3 + 4 timed calls go before
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if "trace_ms" in ln][-1]
    ms = float(line.split("trace_ms")[1].split()[0])
    reused_ms = float(line.split("reused_ms")[1].split()[0])
    print(line)
    raw = np.fromfile(out, np.uint8)
    n = int(raw[:8].view(np.uint64)[0])
    moved = raw[8:8 + 80 * n].view(orc.RAY_DTYPE)
    ray_list = raw[8 + 80 * n:].view(orc.RAY_DTYPE)
    om = orc.Mesh(sc.meshes[0].verts, sc.meshes[0].tris, mesh_mat=sc.meshes[0].material)
    exp = om.trace(rays, sc.m[0], sc.minv[0], sc.normi[0], sc.lights, 0, 7, 8)  # seed 7: the binding passes its call counter, 3 + 4 timed calls went before
    assert n == len(exp) and n > 1_000_000
    assert rays_equal_bits(sort_rays(moved.copy()), sort_rays(exp))
    assert rays_equal_bits(ray_list.copy(), rays)
    # 498 MB cross the link per call (2 M rays in, the updated rayList and ~2 M moved rays out): 8.7 ms at its 57 GB/s.  With moved_rays'
    # capacity kept between calls the call runs at 9.3-9.7 ms; with a vector reserved afresh per call -- the schedulers' own pattern,
    # ImageTracer.h:240 -- the copies also fault in and zero 166 MB of untouched pages (25 ms; any adapter that fills that vector pays it)
    # (a sanity bound, not a measurement: the link's rate differs from box to box -- 9.3-12.1 ms seen over the round; profiles/r05_dropin.txt has the numbers)
    assert 0.5 < reused_ms <= 25.0 and reused_ms <= ms < 80.0
