cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_conv.py tests/test_gpu_norm.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6_tests.txt
python -m pytest tests/test_gpu_golden.py -x -q -m gpu -k "mixed_bf16 or fp16" 2>&1 | tail -15 >> gpurun_out/r6_tests.txt
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --amp > gpurun_out/r6_amp.json 2> gpurun_out/r6_amp.err
DCD_ACT_BF16=0 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --amp > gpurun_out/r6_amp_f32maps.json 2> /dev/null
DCD_PRECISION=bf16 python tools/model_dist_f64.py > gpurun_out/r6_dist.txt 2>&1
bash tools/prof_step.sh r6amp --amp > gpurun_out/r6_prof.log 2>&1
