#!/bin/bash
# per-kernel durations of ONE warm 10 M-triangle acceleration-structure build under rocprofv3 (kernel trace): bash tools/build_stages.sh <tag> [tris]
TAG=${1:-r06}; TRIS=${2:-10000000}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/build_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o trace -- python3 $REPO/tools/build_probe.py tris=$TRIS reps=3 notrace=1 > $OUT/probe.log 2>&1
cd $REPO
cat $OUT/probe.log | grep -v "^\[" | tail -5
python3 - "$OUT" "$TRIS" <<'PY'
import csv, glob, sys
out, tris = sys.argv[1], int(sys.argv[2])
f = glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the LAST build = from the last k_tri_bounds32 on
starts = [i for i, r in enumerate(rows) if "k_tri_bounds32" in r["Kernel_Name"]]
last = rows[starts[-1]:]
t0 = int(last[0]["Start_Timestamp"])
agg = {}
order = []
for r in last:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]
    if "rocprim" in r["Kernel_Name"]:
        n = "rocprim " + ("sort" if ("sort" in r["Kernel_Name"] or "onesweep" in r["Kernel_Name"] or "histogram" in r["Kernel_Name"]) else "scan" if "scan" in r["Kernel_Name"] or "lookback" in r["Kernel_Name"] else "other")
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if n not in agg:
        agg[n] = [0, 0.0]; order.append(n)
    agg[n][0] += 1; agg[n][1] += d
span = (int(last[-1]["End_Timestamp"]) - t0) / 1e3
print("last build of the run: %d kernels, first start to last end %.1f us (%.2f Gtris/s)" % (len(last), span, tris / span / 1e3))
for n in order:
    print("  %-34s x%-3d %8.1f us" % (n[:34], agg[n][0], agg[n][1]))
print("  sum of kernel durations %.1f us; gaps (launch latency, host read-backs) %.1f us" % (sum(v[1] for v in agg.values()), span - sum(v[1] for v in agg.values())))
PY
