cd $GRAFT_REPO_ROOT
for mr in 8 6 4 3; do
  DCD_SWEEP_MIN_ROWS=$mr python tools/time_dcn_layers.py 1 f32 0.5 2>/dev/null | grep -v amdgpu > gpurun_out/r15_b1_mr$mr.txt
  DCD_SWEEP_MIN_ROWS=$mr python tools/time_dcn_layers.py 8 f32 0.5 2>/dev/null | grep TOTAL > gpurun_out/r15_b8_mr$mr.txt
done
