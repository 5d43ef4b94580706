cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_conv.py -x -q -m gpu 2>&1 | grep -E "passed|failed" > gpurun_out/r36_tests.txt
python tools/time_conv.py 8 2>/dev/null | grep "fwd hip" > gpurun_out/r36_conv_f32.txt
DCD_PRECISION=bf16 python tools/time_conv.py 8 2>/dev/null | grep "fwd hip" > gpurun_out/r36_conv_bf16.txt
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-split-line --no-op-line > gpurun_out/r36_f32.json 2>/dev/null
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-split-line --no-op-line --amp > gpurun_out/r36_amp.json 2>/dev/null
