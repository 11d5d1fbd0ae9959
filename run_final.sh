cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err
tail -c 3000 gpurun_out/bench_full.json
timeout 1200 bash profiles/run_profile.sh r01 > gpurun_out/profile_r01.log 2>&1
tail -5 gpurun_out/profile_r01.log
