#!/usr/bin/env python3
"""Condenses gpurun_out/prof_<tag>/ (rocprofv3 csv output) into small, committed summaries under profiles/:
  <tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats per-kernel table (calls, total/avg/min/max ns)
  <tag>_pmc.json           per-kernel PMC sums and per-launch means; HBM traffic with the gfx950 corrections of
                           MI355X_MICROARCH.md (FETCH_SIZE counts 64-B units of 128-B requests for wide loads: reported raw
                           and x2; WRITE_SIZE exact for 16-B stores; both in KiB units -> x1024)
"""
import csv
import glob
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    """(anonymous namespace)::k_closest<true>(RayPlanes, ...) -> k_closest<true>"""
    m = re.search(r"(k_[a-z_0-9]+(<[^>(]*>)?)", name)
    return m.group(1) if m else name.split("(")[0]


def find(pattern):
    return sorted(glob.glob(pattern, recursive=True))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
    base = os.path.join(ROOT, "gpurun_out", "prof_%s" % tag)
    out_stats = os.path.join(ROOT, "profiles", "%s_kernel_stats.csv" % tag)
    stats = find(os.path.join(base, "trace", "**", "*kernel_stats.csv"))
    if stats:
        rows = list(csv.reader(open(stats[0])))
        with open(out_stats, "w", newline="") as f:
            csv.writer(f).writerows(rows)
        print("kernel stats ->", out_stats)
        for r in rows[:12]:
            print("  ", ",".join(r[:8]))
    # per-dispatch durations for our kernels
    traces = find(os.path.join(base, "trace", "**", "*kernel_trace.csv"))
    dur = {}
    if traces:
        for r in csv.DictReader(open(traces[0])):
            k = short(r.get("Kernel_Name", ""))
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            dur.setdefault(k, []).append(d)
    pmc = {}
    regs = {}
    for d in find(os.path.join(base, "pmc_*")):
        for f in find(os.path.join(d, "**", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                k = short(r.get("Kernel_Name", ""))
                e2 = regs.setdefault(k, {})
                for col in ("VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "LDS_Block_Size", "Scratch_Size", "Workgroup_Size", "Grid_Size"):
                    if col in r:
                        e2[col] = r[col]
                c = r.get("Counter_Name", "")
                v = float(r.get("Counter_Value", 0) or 0)
                e = pmc.setdefault(k, {}).setdefault(c, [0.0, 0])
                e[0] += v
                e[1] += 1
    summary = {}
    for k, cs in pmc.items():
        if not any(s in k for s in ("k_trace", "k_closest", "k_any", "k_shade", "k_top", "k_camera", "k_planes", "k_aos", "k_ray_keys", "k_long", "k_fused", "k_round", "k_wave", "k_pack", "k_unpack")):
            continue
        e = {c: {"sum": v[0], "launches": v[1], "per_launch": v[0] / max(1, v[1])} for c, v in cs.items()}
        e["resources"] = regs.get(k, {})
        if k in dur:
            e["avg_duration_ns_trace_pass"] = sum(dur[k]) / len(dur[k])
            e["launches_trace_pass"] = len(dur[k])
        if "FETCH_SIZE" in e:
            raw = e["FETCH_SIZE"]["per_launch"] * 1024.0
            e["hbm_read_bytes_per_launch_raw"] = raw
            e["hbm_read_bytes_per_launch_x2_wide_load_correction"] = raw * 2.0
        if "WRITE_SIZE" in e:
            e["hbm_write_bytes_per_launch"] = e["WRITE_SIZE"]["per_launch"] * 1024.0
        if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e:
            h, m = e["TCC_HIT_sum"]["sum"], e["TCC_MISS_sum"]["sum"]
            e["l2_hit_rate"] = h / max(1.0, h + m)
        summary[k] = e
    # HBM traffic per launch of the two traversal kernels, for bench.py's roofline.traffic (PMC passes are separate runs of the
    # same command; FETCH_SIZE/WRITE_SIZE are in KiB).  The guide's x2 correction (gfx950 tallies a 128-byte fabric request as
    # 64 bytes) was measured for wide streaming reads and leaves other patterns to calibration: profiles/calib_pass.sh +
    # tools/fetch_calib.hip read every 64-byte node of a 2 GiB table exactly once, scattered, with this kernel's four
    # global_load_dwordx4 per lane -- 1.035 requests per node, no 32-byte requests, FETCH_SIZE = 1.035 x the useful bytes, and the
    # pass takes 2.2x the streaming read of the same table (6.0 TB/s if each request moves a 128-byte line, 3.0 TB/s otherwise,
    # against 6.5 TB/s streamed): a node miss moves a whole 128-byte line, so the same x2 applies (profiles/r01_fetch_calibration.txt).
    commit = None
    cf = os.path.join(ROOT, "profiles", ".profiled_commit")  # written by the caller before the tree travels to the GPU box (no .git there)
    if os.path.exists(cf):
        commit = open(cf).read().strip() or None
    src_hash = None
    for f in find(os.path.join(base, "bench_trace.log")) + find(os.path.join(base, "bench_pmc_FETCH_SIZE.log")):  # the hash the PROFILED tree printed in its own bench line
        for line in open(f, errors="replace"):
            if line.startswith("{"):
                src_hash = src_hash or json.loads(line).get("roofline", {}).get("source_hash")
    if src_hash is None:
        try:
            sys.path.insert(0, ROOT)
            from gravit_amd import _build
            src_hash = _build.source_hash()
        except Exception:
            src_hash = None
    traffic = {"tag": tag, "commit": commit, "source_hash": src_hash,
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE over `bench.py --steps 5 --warmup 2`, tag %s" % tag,
               "note": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch; x2 on reads: gfx950 tallies 128-byte fabric requests "
                       "as 64 bytes (MI355X guide), confirmed for this kernel's 64-byte gathers by profiles/r01_fetch_calibration.txt"}
    for k, e in summary.items():
        if k.startswith("k_trace<false") and "hbm_read_bytes_per_launch_raw" in e and "hbm_write_bytes_per_launch" in e:
            traffic["k_closest_bytes_per_launch"] = 2.0 * e["hbm_read_bytes_per_launch_raw"] + e["hbm_write_bytes_per_launch"]
            traffic["k_closest_kernel"] = k
        if k.startswith("k_trace<true") and "hbm_read_bytes_per_launch_raw" in e and "hbm_write_bytes_per_launch" in e:
            traffic["k_any_bytes_per_launch"] = 2.0 * e["hbm_read_bytes_per_launch_raw"] + e["hbm_write_bytes_per_launch"]
            traffic["k_any_kernel"] = k
    if any(k.endswith("_bytes_per_launch") for k in traffic):  # (a pass without k_trace rows must not overwrite the committed file)
        json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1, sort_keys=True)
    out_pmc = os.path.join(ROOT, "profiles", "%s_pmc.json" % tag)
    json.dump(summary, open(out_pmc, "w"), indent=1, sort_keys=True)
    print("pmc ->", out_pmc)
    for k, e in summary.items():
        print(k)
        for c in ("avg_duration_ns_trace_pass", "hbm_read_bytes_per_launch_raw", "hbm_write_bytes_per_launch", "l2_hit_rate"):
            if c in e:
                print("   %-45s %s" % (c, e[c]))
    # bench lines of the passes
    for f in find(os.path.join(base, "bench_*.log")):
        for line in open(f, errors="replace"):
            if line.startswith("{"):
                j = json.loads(line)
                print(os.path.basename(f), "value", round(j["value"], 1), "ms/step", round(j["ms_per_step"], 3), "closest avg ms",
                      round(j["roofline"]["avg_launch_ms"], 4))


if __name__ == "__main__":
    main()
