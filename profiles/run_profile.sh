#!/bin/bash
# Profiles bench.py on the GPU box with rocprofv3 (MI355X guide, section 7): one kernel-trace/stats pass and
# separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit in one pass; never combined with sys/hip traces).
# usage (from the repo root, via gpurun):  bash profiles/run_profile.sh <tag> [bench args...]
# Outputs land in gpurun_out/prof_<tag>/ ; profiles/summarize.py condenses them into profiles/<tag>_*.{csv,json}.
TAG=${1:-r02}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-abi-path $@"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.log 2>&1
for C in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/bench_pmc_$N.log 2>&1
done
cd $REPO
python3 profiles/summarize.py $TAG > $OUT/summary.log 2>&1
tail -30 $OUT/summary.log
