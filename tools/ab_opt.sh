mkdir -p gpurun_out/r3c
for o in "share=1" "share=0" "share=1" "share=0"; do
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-abi-path --opt $o > gpurun_out/r3c/ab.log 2>&1
  python - <<PY
import json
l=[x for x in open("gpurun_out/r3c/ab.log") if x.startswith("{")][-1]; j=json.loads(l); k=j["roofline"]["kernel_ms"]; s=j["steps"]
print("$o: frame %.4f ms  closest %.4f long %.4f any %.4f" % (j["ms_per_step"], k["ms_closest"]/s, k["ms_long"]/s, k["ms_any"]/s))
PY
done
