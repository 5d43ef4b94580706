cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_heads.py tests/test_gpu_golden.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r10_tests.txt
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-split-line --no-op-line > gpurun_out/r10_f32.json 2> gpurun_out/r10_f32.err
DCD_EDGE_BRANCH_GEMM=0 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-split-line --no-op-line > gpurun_out/r10_f32_off.json 2> /dev/null
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-split-line --no-op-line --batch 1 > gpurun_out/r10_b1.json 2> /dev/null
DCD_EDGE_BRANCH_GEMM=0 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-split-line --no-op-line --batch 1 > gpurun_out/r10_b1_off.json 2> /dev/null
