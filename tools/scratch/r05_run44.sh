cd $GRAFT_REPO_ROOT
PREC=bf16x3 N=20 python tools/scratch/graph_lr0.py 2>&1 | grep "^replay" | tail -4 > gpurun_out/r44_lr0_x3.txt
PREC=f32 N=20 python tools/scratch/graph_lr0.py 2>&1 | grep "^replay" | tail -3 > gpurun_out/r44_lr0_f32.txt
NB=1 N=81 MODES=eager,graph python tools/scratch/graph_vs_eager.py 2>&1 | grep "^step" > gpurun_out/r44_gve.txt
python -m pytest tests/test_gpu_heads.py tests/test_gpu_golden.py -q -m gpu 2>&1 | grep -E "passed|failed" > gpurun_out/r44_tests.txt
