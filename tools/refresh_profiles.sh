#!/bin/bash
# Regenerates the measured text files under profiles/ on the GPU box (outputs in gpurun_out/refresh_<tag>/; copy them over afterwards):
#   bash tools/refresh_profiles.sh <tag>
TAG=${1:-r05}
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/refresh_$TAG
mkdir -p $OUT
cd $REPO
python3 bench.py --steps 20 --warmup 3 > $OUT/bench.log 2>&1 && grep '^{' $OUT/bench.log | tail -1 > $OUT/bench.json
python3 tools/tail_probe.py any > $OUT/tail_fit.txt 2>&1
python3 tools/bench_configs.py roofline=1 > $OUT/configs.txt 2>&1
python3 bench.py --domains 8 --steps 20 --warmup 3 > $OUT/domains8.log 2>&1
for n in 2 4 8; do python3 bench.py --inproc-ranks $n --steps 10 --warmup 2 > $OUT/inproc_$n.log 2>&1; done
for n in 2 4 8; do python3 bench.py --inproc-ranks $n --steps 10 --warmup 2 --no-extra-legs --opt skip_known=1 > $OUT/inproc_s_$n.log 2>&1; done   # the opt-in known-miss shortcut
python3 tools/tick_probe.py > $OUT/tick_probe.txt 2>&1; python3 tools/tick_probe.py frame_timing=1 >> $OUT/tick_probe.txt 2>&1
python3 tools/tick_probe.py inline_kb=0 >> $OUT/tick_probe.txt 2>&1; python3 tools/tick_probe.py inline_kb=1024 >> $OUT/tick_probe.txt 2>&1; python3 tools/tick_probe.py bsp=1 >> $OUT/tick_probe.txt 2>&1
bash tools/dropin_probe.sh > $OUT/dropin.txt 2>&1
bash tools/timeline.sh $TAG > $OUT/timeline.txt 2>&1
tail -3 $OUT/tail_fit.txt; grep rounds $OUT/configs.txt
