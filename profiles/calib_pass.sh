#!/bin/bash
# FETCH_SIZE calibration for the gather pattern: bash profiles/calib_pass.sh   (via gpurun, from the repo root)
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/calib
mkdir -p $OUT
# the tool is built here from its source (no binaries in the tree)
hipcc -O3 --offload-arch=gfx950 -o $REPO/tools/fetch_calib $REPO/tools/fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
timeout 120 $REPO/tools/fetch_calib 25 > $OUT/plain.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o pmc -- $REPO/tools/fetch_calib 25 > $OUT/fetch.log 2>&1
timeout 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/rdreq -o pmc -- $REPO/tools/fetch_calib 25 > $OUT/rdreq.log 2>&1
cat $OUT/plain.log
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = "gather_once" if "gather_once" in r["Kernel_Name"] else ("stream" if "k_stream" in r["Kernel_Name"] else None)
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(k, {c: sum(x) / len(x) for c, x in v.items()})
PY
