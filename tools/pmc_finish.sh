#!/bin/bash
# counters of k_finish on the benchmark's --domains 8 frame (one rank, 8 tiles of the 10 M soup):  bash tools/pmc_finish.sh <tag> [opt=value ...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmcf_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAVES SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "TCC_HIT_sum TCC_MISS_sum" FETCH_SIZE; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $OUT/p$i -o pmc -- python3 $REPO/tools/frame_timeline.py run domains=8 "$@" > $OUT/p$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        for key in ("k_finish", "k_trace", "k_long_closest"):
            if key + "<" in n or key + "(" in n:
                acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k, {c: "%.4g (x%d)" % (sum(v) / len(v), len(v)) for c, v in sorted(d.items())})
PY
rm -rf $OUT/p*/
