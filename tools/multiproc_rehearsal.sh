export GVT_HIP_RCCL_LIB=$PWD/tests/fake_rccl/libfakerccl.so TMPDIR=/tmp/mp_$$ MASTER_ADDR=127.0.0.1 MASTER_PORT=29611 WORLD_SIZE=3
mkdir -p $TMPDIR
for r in 0 1 2; do RANK=$r LOCAL_RANK=$r python tests/multiproc_worker.py gpurun_out/r05_multiproc_verdict_3.json > gpurun_out/r05_multiproc_rank$r.log 2>&1 & done
wait
cat gpurun_out/r05_multiproc_verdict_3.json
