#!/bin/bash
# kernel timeline of one benchmark frame: bash tools/timeline.sh <tag> [opt=value ...]
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/tl_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $OUT -o tl -- python3 $REPO/tools/frame_timeline.py run "$@" > $OUT/run.log 2>&1
cd $REPO && python3 tools/frame_timeline.py parse $OUT | tee $OUT/timeline.txt
