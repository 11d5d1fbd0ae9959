#!/bin/bash
# One rocprofv3 counter pass over tools/launch_probe.py:  bash tools/pmc.sh <tag> "<COUNTER ...>" [opt=value ...]
# (separate passes per counter group, never combined with trace domains; the program itself follows `--`)
TAG=$1; CNT=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --pmc $CNT --output-format csv -d $OUT -o pmc -- python3 $REPO/tools/launch_probe.py "$@" > $OUT/probe.log 2>&1
echo "rocprofv3 rc=$?" >> $OUT/probe.log
python3 - "$OUT" <<'PY'
import csv, glob, re, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        m = re.search(r"(k_[a-z_0-9]+(<[^>(]*>)?)", r["Kernel_Name"])
        if not m: continue
        e = acc[m.group(1)][r["Counter_Name"]]
        e[0] += float(r["Counter_Value"]); e[1] += 1
for k in sorted(acc):
    if not k.startswith(("k_trace", "k_long")): continue
    print(k, {c: "%.5g" % (v[0] / max(1, v[1])) for c, v in sorted(acc[k].items())}, "launches", max(v[1] for v in acc[k].values()))
PY
tail -2 $OUT/probe.log
