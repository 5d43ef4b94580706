cd $GRAFT_REPO_ROOT
DCD_STEP_GRAPH=1 bash tools/prof_step.sh graphb8b > gpurun_out/r61_prof.log 2>&1
grep -c fillBuffer gpurun_out/r05_step_graphb8b_sequence.txt > gpurun_out/r61_memsets.txt
PREC=f32 python tools/scratch/graph_twin2.py 2>&1 | grep "^step" > gpurun_out/r61_twin2_f32.txt
NB=1 N=81 MODES=eager,graph python tools/scratch/graph_vs_eager.py 2>&1 | grep "^step" > gpurun_out/r61_gve.txt
python -m pytest tests/test_gpu_trunk_moments.py tests/test_gpu_golden.py tests/test_gpu_heads.py -q -m gpu 2>&1 | grep -E "passed|failed" > gpurun_out/r61_tests.txt
