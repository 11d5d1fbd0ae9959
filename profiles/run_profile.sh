#!/bin/bash
# Profiles bench.py on the GPU box with rocprofv3 (MI355X guide, section 7): one kernel-trace/stats pass and
# separate PMC passes (FETCH_SIZE and WRITE_SIZE do not fit in one pass; never combined with sys/hip traces).
# usage (from the repo root, via gpurun):  bash profiles/run_profile.sh <tag> [bench args...]
# Outputs land in gpurun_out/prof_<tag>/ ; profiles/summarize.py condenses them into profiles/<tag>_*.{csv,json}.
TAG=${1:-r04}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 5 --warmup 2 --no-cpu-baseline --no-abi-path --no-sustained $@"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py $ARGS > $OUT/bench_trace.log 2>&1
# (TA_* counters abort rocprofv3 on this pool -- signal 6, the pass then runs into its timeout: gpurun_out/pmc_q0_ta, round 3 -- so the
# vector-memory path is read through the TCP counters: tag look-ups, requests to L2, latencies, stall cycles)
for C in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/pmc_$N -o pmc -- python3 $REPO/bench.py $ARGS > $OUT/bench_pmc_$N.log 2>&1
done
cd $REPO
python3 profiles/summarize.py $TAG > $OUT/summary.log 2>&1
tail -30 $OUT/summary.log
# the raw counter dumps stay on the box (gpurun copies at most 64 MiB back): the summaries are profiles/<tag>_kernel_stats.csv, <tag>_pmc.json, traffic.json
mkdir -p $OUT/keep && cp $REPO/profiles/${TAG}_kernel_stats.csv $REPO/profiles/${TAG}_pmc.json $REPO/profiles/traffic.json $OUT/keep/ 2>/dev/null
rm -rf $OUT/pmc_* $OUT/trace
