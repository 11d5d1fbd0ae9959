#!/bin/bash
# One rocprofv3 PMC pass over bench.py: bash profiles/pmc_pass.sh <tag> "<COUNTER ...>" [bench args]
TAG=$1; CNT=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc $CNT --output-format csv -d $OUT -o pmc -- python3 $REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-abi-path "$@" > $OUT/bench.log 2>&1
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
    if not k.startswith(("k_trace", "k_closest", "k_any", "k_shade", "k_top")): continue
    print(k, {c: "%.4g" % (v[0] / max(1, v[1])) for c, v in sorted(acc[k].items())})
PY
