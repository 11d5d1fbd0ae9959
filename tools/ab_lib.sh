#!/bin/bash
# A/B of library BUILDS on the benchmark frame: every build (a path to a libgvt_hip*.so, optionally "path:opt=v,opt=v") is run `reps` times, alternating, through
# bench.py's per-kernel HIP events.   bash tools/ab_lib.sh reps lib1[:opts] lib2[:opts] ...
reps=$1; shift
mkdir -p gpurun_out/ab
for r in $(seq 1 $reps); do
  for spec in "$@"; do
    lib=${spec%%:*}; opts=""; [ "$lib" != "$spec" ] && opts=${spec#*:}
    args=""; for kv in ${opts//,/ }; do args="$args --opt $kv"; done
    GVT_HIP_LIB=$PWD/$lib python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-abi-path --no-sustained $args > gpurun_out/ab/ab.log 2>&1
    python - "$spec" <<'PY'
import json, sys
l=[x for x in open("gpurun_out/ab/ab.log") if x.startswith("{")]
if not l:
    print("%-60s FAILED: %s" % (sys.argv[1], open("gpurun_out/ab/ab.log").read()[-400:])); sys.exit(0)
j=json.loads(l[-1]); k=j["roofline"]["kernel_ms"]; s=j["steps"]
print("%-60s frame %.4f ms  closest %.4f long %.4f any %.4f  value %.0f" % (sys.argv[1], j["ms_per_step"], k["ms_closest"]/s, k["ms_long"]/s, k["ms_any"]/s, j["value"]))
PY
  done
done
