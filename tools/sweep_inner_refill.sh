mkdir -p gpurun_out/r3b
for im in 8 16 24 32; do for rm in 8 16 24 32; do
  echo "inner_min=$im refill_min=$rm: $(python3 tools/launch_probe.py reps=5 inner_min=$im refill_min=$rm | tail -1)"
done; done > gpurun_out/r3b/sweep_im_rm.log 2>&1
cat gpurun_out/r3b/sweep_im_rm.log
