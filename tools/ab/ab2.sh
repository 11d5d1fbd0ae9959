bash tools/ab_lib.sh 3 tools/ab/libgvt_old.so tools/ab/libgvt_new.so tools/ab/libgvt_new5.so tools/ab/libgvt_new_exp.so > gpurun_out/r06_ab_diet2.txt 2>&1
cat gpurun_out/r06_ab_diet2.txt
for l in old new5 old new5; do GVT_HIP_LIB=$PWD/tools/ab/libgvt_$l.so python bench.py --domains 8 --steps 40 --warmup 5 --no-cpu-baseline --no-abi-path --no-sustained 2>/dev/null | python -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$l domains8 %.4f ms' % j['ms_per_step'])"; done
