for l in old new6 old new6; do GVT_HIP_LIB=$PWD/tools/ab/libgvt_$l.so python bench.py --domains 8 --steps 40 --warmup 5 --no-cpu-baseline --no-abi-path --no-sustained 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$l domains8 %.4f ms' % j['ms_per_step'])"; done
for l in old new6 old new6; do echo "== $l"; GVT_HIP_LIB=$PWD/tools/ab/libgvt_$l.so python tools/bench_configs.py noref=1 frames=30 2>&1 | grep -i "rounds\|ms" | head -12; done
bash tools/ab_lib.sh 2 tools/ab/libgvt_old.so tools/ab/libgvt_new6.so
