cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_conv.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r9_tests.txt
python tools/scratch/conv_bf16_probe.py > gpurun_out/r9_probe.txt 2>&1
DCD_CONV_BF16_PIPE=0 DCD_PROBE_ONLY=1 python tools/scratch/conv_bf16_probe.py > gpurun_out/r9_probe_old.txt 2>&1
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --amp > gpurun_out/r9_amp.json 2> gpurun_out/r9_amp.err
