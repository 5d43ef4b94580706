cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_trunk_moments.py tests/test_gpu_golden.py tests/test_gpu_heads.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r7_tests.txt
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --amp > gpurun_out/r7_amp.json 2> gpurun_out/r7_amp.err
DCD_TRUNK_GRAM_BF16=0 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --amp > gpurun_out/r7_amp_f32gram.json 2> /dev/null
