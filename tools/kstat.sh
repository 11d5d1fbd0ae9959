#!/bin/bash
# per-kernel average durations of a short bench.py run under rocprofv3 (kernel trace + stats):  bash tools/kstat.sh <tag> [bench args...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/kstat_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-abi-path --no-sustained "$@" > $OUT/bench.log 2>&1
cd $REPO
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("k_shade", "k_trace", "k_long", "k_cam1")):
        print("$TAG %-60s calls %5s avg %9.1f us" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
