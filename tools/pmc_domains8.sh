REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/pmc_d8
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for C in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY"; do
  N=$(echo $C | tr ' ' '_' | cut -c1-30)
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $OUT/$N -o pmc -- python3 $REPO/bench.py --domains 8 --steps 3 --warmup 1 --no-cpu-baseline --no-abi-path > $OUT/bench_$N.log 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 $REPO/bench.py --domains 8 --steps 5 --warmup 1 --no-cpu-baseline --no-abi-path > $OUT/bench_trace.log 2>&1
ls $OUT
