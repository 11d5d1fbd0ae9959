#!/bin/bash
# A/B of library options on the benchmark frame (per-kernel HIP events of bench.py): bash tools/ab_opt.sh "opt=v[,opt=v]" ["opt=v" ...]
mkdir -p gpurun_out/ab
for o in "$@"; do
  args=""; for kv in ${o//,/ }; do args="$args --opt $kv"; done
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-abi-path --no-sustained $args > gpurun_out/ab/ab.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/ab/ab.log") if x.startswith("{")][-1]; j=json.loads(l); k=j["roofline"]["kernel_ms"]; s=j["steps"]
print("%-40s frame %.4f ms  closest %.4f long %.4f any %.4f" % ("$o", j["ms_per_step"], k["ms_closest"]/s, k["ms_long"]/s, k["ms_any"]/s))
PY
done
