# finish_rays (rounds of at most this many rays go through k_finish, one launch, a wave per ray) across the multi-instance workloads
cd ${GRAFT_REPO_ROOT:-.}
for f in 32768 8192 4096 2048 1024 0; do
  echo "== finish_rays=$f"
  python3 tools/bench_configs.py only=4,5 finish_rays=$f 2>&1 | grep rounds | cut -c1-150
  python3 bench.py --domains 8 --steps 10 --warmup 2 --opt finish_rays=$f --no-cpu-baseline --no-abi-path --no-sustained 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('soup in 8 domains, one rank: %.3f ms/frame' % j['ms_per_step'])"
  python3 bench.py --inproc-ranks 4 --steps 6 --warmup 2 --opt finish_rays=$f --no-extra-legs 2>/dev/null | python3 -c "
import json,sys
j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); v=j['variants']['domain_async']; print('soup, 4 in-process ranks async: %.3f ms/frame, %.3f ms/tick' % (v['ms_per_step'], v['ms_per_tick']))"
done
