"""Copies what tools/refresh_profiles.sh measured on the GPU box (gpurun_out/refresh_<tag>/) into the tracked profiles/<tag>_* files.
   usage: python tools/collect_profiles.py <tag>"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
src = os.path.join(ROOT, "gpurun_out", "refresh_" + tag)
dst = os.path.join(ROOT, "profiles")
commit = open(os.path.join(dst, ".profiled_commit")).read().strip()


def last_json(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


bench = json.load(open(os.path.join(src, "bench.json")))
h = bench["roofline"]["source_hash"]
json.dump(bench, open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1, sort_keys=True)
open(os.path.join(dst, tag + "_tail_fit.txt"), "w").write("# tools/tail_probe.py at %s (source hash %s): the two traversal launches of the benchmark scene for 0.25 M .. 4.2 M camera rays\n" % (commit, h) + open(os.path.join(src, "tail_fit.txt")).read())
open(os.path.join(dst, tag + "_configs.txt"), "w").write("# tools/bench_configs.py at %s: BASELINE.json's configurations on one MI355X, 10 frames each\n" % commit + open(os.path.join(src, "configs.txt")).read())
tl = [l for l in open(os.path.join(src, "timeline.txt")).read().splitlines() if l.strip()]
open(os.path.join(dst, tag + "_frame_timeline.txt"), "w").write(
    "# tools/timeline.sh (rocprofv3 --kernel-trace of one benchmark frame, native tracer, 10 M soup, 1080p) at %s: start, gap to the previous kernel's end, duration\n" % commit + "\n".join(tl) +
    "\n# (two slices of this frame on two streams, and four: profiles/%s_frame_slices.txt -- measured slower, removed)\n" % tag)
rows, legs = [], []
for n in (2, 4, 8):
    j = last_json(os.path.join(src, "inproc_%d.log" % n))
    for name, v in j["variants"].items():
        ph = v["phase_ms_per_step_max_over_ranks"]
        rows.append("%d %-18s %8.3f %6.1f %7.1f %6.1f %10d %11d | %5.2f %8.2f %7.2f %9.2f %9.2f" % (
            n, name, v["ms_per_step"], v.get("ticks_per_step", 0), v.get("launch_chains_per_step", 0), v.get("host_syncs_per_step", 0),
            v.get("rays_sent_per_step", 0), v.get("bytes_sent_per_step", 0), ph["chain"], ph["announce"], ph["payload"], ph["composite"], ph["host_wait"]))
    ps = os.path.join(src, "inproc_s_%d.log" % n)
    if os.path.exists(ps):  # the same with the opt-in known-miss shortcut (skip_known = 1)
        for name, v in last_json(ps)["variants"].items():
            if name == "image_replicated":
                continue
            ph = v["phase_ms_per_step_max_over_ranks"]
            rows.append("%d %-18s %8.3f %6.1f %7.1f %6.1f %10d %11d | %5.2f %8.2f %7.2f %9.2f %9.2f" % (
                n, name + "+skip", v["ms_per_step"], v.get("ticks_per_step", 0), v.get("launch_chains_per_step", 0), v.get("host_syncs_per_step", 0),
                v.get("rays_sent_per_step", 0), v.get("bytes_sent_per_step", 0), ph["chain"], ph["announce"], ph["payload"], ph["composite"], ph["host_wait"]))
    c4, wk = j.get("config4_bunny_grid"), j.get("weak_soup")
    if c4:
        for name in ("domain_async", "domain_bsp"):
            v = c4[name]
            legs.append("%d config4_bunny_grid %-13s %8.3f ms/frame %8.1f Mrays/s  ticks %.1f  rays sent %d  rays %d" % (n, name, v["ms_per_step"], v["value"], v["ticks_per_step"], v["rays_sent_per_step"], v["rays_per_step"]))
    if wk:
        legs.append("%d weak_soup (%d tiles x %d triangles, film %dx%d)  %8.3f ms/frame %8.1f Mrays/s  ticks %.1f  rays sent %d  rays %d; per-rank roofline frac of the dominant kernel: %s" % (
            n, wk["tiles"], wk["tris_per_tile"], wk["film"][0], wk["film"][1], wk["ms_per_step"], wk["value"], wk["ticks_per_step"], wk["rays_sent_per_step"], wk["rays_per_step"],
            " ".join("%.3f" % r["frac"] for r in wk["roofline_per_rank"])))
d8 = last_json(os.path.join(src, "domains8.log"))
open(os.path.join(dst, tag + "_domain_ticks.txt"), "w").write(
    "# bench.py --inproc-ranks N --steps 10 --warmup 2 at %s (source hash %s): the native multi-rank frame loop with N in-process ranks sharing ONE MI355X (hub transport).\n"
    "# Tick counts, rays / bytes sent and the per-phase times (max over ranks, ms per frame; frame_timing on: five more event calls per tick, and no speculative tick parts -- spec_ticks needs it off) of the config-3 soup cut into N x-y tiles;\n"
    "# NOT a scaling number: the ranks' launch chains serialise on one device.  Default = the reference's hop-by-hop shuffle rule; rows '+skip': the opt-in known-miss shortcut\n"
    "# (skip_known = 1: fewer hand-back hops between overlapping tiles, not image-identical in general -- DESIGN 6).  Payloads of at most inline_kb = 16 KiB per pair ride inside the announce.\n"
    "# N variant            ms/frame  ticks  chains  syncs  rays_sent  bytes_sent | chain announce payload composite host_wait\n" % (commit, h) + "\n".join(rows) +
    "\n# the extra legs of the same invocation (BASELINE configs[3] at its 1900x1080 film; the weak-scaling soup):\n" + "\n".join(legs) +
    "\n# bench.py --domains 8 (one rank owns all 8 tiles): %.3f ms per frame, %s launch chains, %s host synchronisations\n"
    % (d8["ms_per_step"], d8["config"].get("launch_chains_per_step"), d8["config"].get("host_syncs_per_step")))
dp = [l for l in open(os.path.join(src, "dropin.txt")) if l.startswith("dropin_demo: trace_ms")]
open(os.path.join(dst, tag + "_dropin.txt"), "w").write(
    "# tools/dropin_probe.sh at %s: oracle/_ref/dropin_demo = gravit_amd/host/HipMeshAdapter.cpp compiled against the reference's own headers; 2,073,600 gvt::render::actor::Ray\n"
    "# (1920x1080 camera rays of the bunny scene) through gvt::render::Adapter::trace -- 166 MB in, 166 MB of updated rayList and 166 MB of moved rays out = 498 MB over a 57 GB/s link (8.7 ms).\n"
    "# trace_ms: moved_rays reserved afresh per call (ImageTracer.h:240), its pages untouched: the copies fault them in and zero them; reused_ms: its capacity kept between calls.\n" % commit + "".join(dp) +
    "# bench.py abi_path (gvt_hip_trace on the benchmark frame's 1,040,400 rays, 250 MB): %.2f ms with write-back (%.1f GB/s), %.2f ms without (%.1f GB/s)\n" % (
        bench["abi_path"]["ms"], bench["abi_path"]["pcie_GB/s"], bench["abi_path"]["no_write_back"]["ms"], bench["abi_path"]["no_write_back"]["pcie_GB/s"]))
print("profiles/%s_* refreshed from %s (commit %s, source hash %s)" % (tag, src, commit, h))
