#!/bin/bash
# kernel / copy / HIP-API timeline of the toy two-rank frame (tools/tick_probe.py): bash tools/tick_timeline.sh <tag> [opt=value ...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/tick_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $OUT -o tk -- python3 $REPO/tools/tick_probe.py frames=30 "$@" > $OUT/run.log 2>&1
cd $REPO && python3 - $OUT <<'PY'
import csv, glob, os, re, sys
out = sys.argv[1]
def load(pat):
    f = glob.glob(os.path.join(out, "**", pat), recursive=True)
    return list(csv.DictReader(open(f[0]))) if f else []
ker, cpy, api = load("*kernel_trace.csv"), load("*memory_copy_trace.csv"), load("*hip_api_trace.csv")
ev = []
for r in ker:
    m = re.search(r"(k_[a-z_0-9]+)", r["Kernel_Name"]); ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + (m.group(1) if m else r["Kernel_Name"][:40]), r.get("Queue_Id", "")))
for r in cpy:
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " + r.get("Bytes", ""), ""))
ev.sort()
# last but one frame: from a k_zero_totals to the next one
z = [i for i, e in enumerate(ev) if "k_zero_totals" in e[2]]
a, b = z[-4], z[-2]   # two ranks: two k_zero_totals per frame
t0 = ev[a][0]
for s, e, n, q in ev[a:b]:
    print("%9.1f us %7.1f us  %s %s" % ((s - t0) / 1e3, (e - s) / 1e3, n, q))
print("GPU events in the frame: %d, span %.1f us" % (b - a, (ev[b][0] - t0) / 1e3))
if api:
    lo, hi = t0, ev[b][0]
    cnt = {}
    for r in api:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s >= lo and s < hi:
            k = r["Function"]; c = cnt.setdefault(k, [0, 0]); c[0] += 1; c[1] += e - s
    print("HIP API calls inside that span (both ranks): " + ", ".join("%s x%d %.0f us" % (k, v[0], v[1] / 1e3) for k, v in sorted(cnt.items(), key=lambda kv: -kv[1][1])))
PY
