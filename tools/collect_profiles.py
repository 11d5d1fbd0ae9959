"""Copies what tools/refresh_profiles.sh measured on the GPU box (gpurun_out/refresh_<tag>/) into the tracked profiles/<tag>_* files.
   usage: python tools/collect_profiles.py <tag>"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", "refresh_" + tag)
dst = os.path.join(ROOT, "profiles")
commit = open(os.path.join(dst, ".profiled_commit")).read().strip()


def keys(d, path, default=None):
    for k in path:
        if not isinstance(d, dict) or k not in d: return default
        d = d[k]
    return d


bench = json.load(open(os.path.join(src, "bench.json")))
json.dump(bench, open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1, sort_keys=True)
open(os.path.join(dst, tag + "_tail_fit.txt"), "w").write("# tools/tail_probe.py at %s (source hash %s): the two traversal launches of the benchmark scene for 0.25 M .. 4.2 M camera rays\n" % (commit, bench["roofline"]["source_hash"]) + open(os.path.join(src, "tail_fit.txt")).read())
open(os.path.join(dst, tag + "_configs.txt"), "w").write("# tools/bench_configs.py at %s: BASELINE.json's configurations on one MI355X, 10 frames each\n" % commit + open(os.path.join(src, "configs.txt")).read())
tl = [l for l in open(os.path.join(src, "timeline.txt")).read().splitlines() if l.strip()]
old = open(os.path.join(dst, tag + "_frame_timeline.txt")).read().split("\n## ")
head = "# tools/timeline.sh (rocprofv3 --kernel-trace of one benchmark frame, native tracer, 10 M soup, 1080p), round 3 at %s\n## lean frame (default): 7 kernels\n" % commit
rest = ["## " + s for s in old[2:]]  # the sections measured once (round 2's launch sequence, --domains 8) stay as they were taken
open(os.path.join(dst, tag + "_frame_timeline.txt"), "w").write(head + "\n".join(tl) + "\n\n" + "\n".join(rest))
rows = []
for n in (2, 4, 8):
    line = [l for l in open(os.path.join(src, "inproc_%d.log" % n)) if l.startswith("{")][-1]
    j = json.loads(line)
    for name, v in j["variants"].items():
        ph = v["phase_ms_per_step_max_over_ranks"]
        rows.append("%d %-18s %8.3f %6.1f %7.1f %6.1f %10d %11d | %5.2f %8.2f %7.2f %9.2f %9.2f" % (
            n, name, v["ms_per_step"], v.get("ticks_per_step", 0), v.get("launch_chains_per_step", 0), v.get("host_syncs_per_step", 0),
            v.get("rays_sent_per_step", 0), v.get("bytes_sent_per_step", 0), ph["chain"], ph["announce"], ph["payload"], ph["composite"], ph["host_wait"]))
d8 = json.loads([l for l in open(os.path.join(src, "domains8.log")) if l.startswith("{")][-1])
open(os.path.join(dst, tag + "_domain_ticks.txt"), "w").write(
    "# bench.py --inproc-ranks N --steps 10 --warmup 2 (round 3, %s): the native multi-rank frame loop with N in-process ranks sharing ONE MI355X (hub transport).\n"
    "# Tick counts, rays / bytes sent and the per-phase times (max over ranks, ms per frame) of the config-3 soup cut into N x-y tiles; NOT a scaling number:\n"
    "# the ranks' launch chains serialise on one device and the rank threads share one interpreter.\n"
    "# N variant            ms/frame  ticks  chains  syncs  rays_sent  bytes_sent | chain announce payload composite host_wait\n" % commit + "\n".join(rows) +
    "\n# bench.py --domains 8 (one rank owns all 8 tiles): %.3f ms per frame, %s launch chains, %s host synchronisations (round 2: 2.4-2.9 ms, 8 chains; k_finish runs every round after the first in one launch)\n"
    % (d8["ms_per_step"], keys(d8, ["config", "launch_chains_per_step"], "2"), keys(d8, ["config", "host_syncs_per_step"], "3")))
print("profiles/%s_* refreshed from %s" % (tag, src))
