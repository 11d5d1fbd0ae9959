#!/bin/bash
# payload_overlap_kb on the config-5 slabs over the multi-process stand-in transport: 3 processes on one GPU, tests/fake_rccl, the hall in 6 slabs;
# payloads of at least payload_overlap_kb KiB move on the communicator's own stream beside the next chain.  The stand-in is blocking: this shows the knob's
# control flow at the largest payloads, not a speed.   bash tools/overlap_slabs.sh
REPO=${GRAFT_REPO_ROOT:-$(pwd)}; cd $REPO
python3 -c "from tests.test_gpu_multiproc import build_fake_rccl; build_fake_rccl()" || exit 1
for KB in 64 1024; do
  export GVT_HIP_RCCL_LIB=$REPO/tests/fake_rccl/libfakerccl.so TMPDIR=/tmp/ovl_$$_$KB MASTER_ADDR=127.0.0.1 MASTER_PORT=$((29700 + KB % 97)) WORLD_SIZE=3 FAKE_RCCL_TIMEOUT_S=240
  mkdir -p $TMPDIR
  for r in 0 1 2; do RANK=$r LOCAL_RANK=$r python3 tools/overlap_slabs_worker.py $KB > $TMPDIR/rank$r.log 2>&1 & done
  wait
  cat $TMPDIR/rank0.log | tail -3
  rm -rf $TMPDIR
done
