#!/bin/bash
# rocprofv3 counter passes over ONE configuration of tools/bench_configs.py (counters for the configurations packets and child
# ordering were judged on):  bash tools/pmc_configs.sh <config number> <tag>   -> gpurun_out/pmc_cfg<N>/, summary profiles/<tag>_pmc_cfg<N>.json
# Separate passes per counter group (FETCH_SIZE and WRITE_SIZE do not share a pass), never combined with a trace domain; the program follows `--`.
CFG=$1; TAG=${2:-r05}; shift; shift   # further arguments: opt=value for tools/bench_configs.py (GVT_HIP_LIB in the environment selects the library)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_${TAG}_cfg$CFG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="only=$CFG frames=5 noref=1 $@"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/tools/bench_configs.py $ARGS roofline=1 > $OUT/trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES" \
         "TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout -k 10 240 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- python3 $REPO/tools/bench_configs.py $ARGS > $OUT/pmc_$N.log 2>&1
  echo "pass $N rc=$?"
done
cd $REPO
python3 - $OUT $CFG $TAG <<'PY'
import csv, glob, json, os, re, sys
out, cfg, tag = sys.argv[1], sys.argv[2], sys.argv[3]
root = os.path.dirname(os.path.dirname(os.path.abspath(out)))
def short(name):
    m = re.search(r"(k_[a-z_0-9]+(<[^>(]*>)?)", name)
    return m.group(1) if m else name.split("(")[0]
dur = {}
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur.setdefault(short(r["Kernel_Name"]), []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
pmc = {}
for f in glob.glob(out + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        e = pmc.setdefault(short(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], [0.0, 0])
        e[0] += float(r["Counter_Value"] or 0); e[1] += 1
res = {}
for k, cs in pmc.items():
    if not k.startswith(("k_trace", "k_packet", "k_long", "k_shade", "k_wave_any", "k_frame1", "k_finish")): continue
    e = {c: v[0] / max(1, v[1]) for c, v in cs.items()}   # per launch
    e["launches_counted"] = max(v[1] for v in cs.values())
    if k in dur: e["avg_launch_us_trace_pass"] = sum(dur[k]) / len(dur[k]) / 1e3; e["launches_trace_pass"] = len(dur[k])
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e: e["hbm_bytes_per_launch"] = (2.0 * e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0
    if e.get("SQ_INSTS_VMEM_RD") and "TCP_TCP_LATENCY_sum" in e: e["cycles_a_load_instruction_spends_in_L1"] = e["TCP_TCP_LATENCY_sum"] / e["SQ_INSTS_VMEM_RD"]
    if e.get("SQ_WAVE_CYCLES"): e["wait_any_fraction"] = e.get("SQ_WAIT_ANY", 0) / e["SQ_WAVE_CYCLES"]; e["active_inst_fraction"] = e.get("SQ_ACTIVE_INST_ANY", 0) / e["SQ_WAVE_CYCLES"]
    if "TCC_HIT_sum" in e and "TCC_MISS_sum" in e: e["l2_hit_rate"] = e["TCC_HIT_sum"] / max(1.0, e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
    res[k] = e
line = [l for l in open(os.path.join(out, "trace.log"), errors="replace") if l.startswith(("config", "    kernel classes"))]
sys.path.insert(0, root)
try:
    from gravit_amd import _build
    h = _build.source_hash()
except Exception:
    h = None
json.dump({"config": cfg, "source_hash": h, "command": "rocprofv3 --pmc <group> -- python3 tools/bench_configs.py only=%s frames=5 noref=1 (one pass per group); per-launch means; "
           "hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 corrections as in profiles/summarize.py)" % cfg,
           "bench_lines_of_the_trace_pass": [l.rstrip() for l in line], "kernels": res}, open(os.path.join(root, "profiles", "%s_pmc_cfg%s.json" % (tag, cfg)), "w"), indent=1, sort_keys=True)
for k, e in sorted(res.items()):
    print(k, {c: ("%.4g" % v) for c, v in e.items() if c in ("avg_launch_us_trace_pass", "hbm_bytes_per_launch", "cycles_a_load_instruction_spends_in_L1", "wait_any_fraction", "l2_hit_rate", "launches_counted")})
PY
rm -rf $OUT/pmc_* $OUT/trace   # the raw counter dumps (tens of MiB) stay on the box: gpurun copies at most 64 MiB back; the summary is profiles/<tag>_pmc_cfg<N>.json, copied next
cp $REPO/profiles/${TAG}_pmc_cfg$CFG.json $OUT/ 2>/dev/null
