#!/bin/bash
# round 5, GPU run 1: the bf16 one-product kernels -- parity tests, model distances, conv timings, bench lines
mkdir -p gpurun_out/r05
cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_dcn.py tests/test_gpu_conv.py -x -q -k "bf16 or split or policy" 2>&1 | tail -15) > gpurun_out/r05/t1_ops.txt
(timeout 900 python -m pytest tests/test_gpu_golden.py -x -q -k "fp16 or mixed_bf16 or baseline_size" 2>&1 | tail -15) > gpurun_out/r05/t1_model.txt
DCD_PRECISION=bf16 timeout 600 python tools/model_dist_f64.py > gpurun_out/r05/dist_bf16.txt 2>&1
timeout 600 python tools/time_conv.py > gpurun_out/r05/conv_f32.txt 2>&1
DCD_PRECISION=bf16 timeout 600 python tools/time_conv.py > gpurun_out/r05/conv_bf16.txt 2>&1
DCD_PRECISION=bf16x3 DCD_CONV_SPLIT_MIN_MAP=0 timeout 600 python tools/time_conv.py > gpurun_out/r05/conv_bf16x3.txt 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r05/bench_default.json 2> gpurun_out/r05/bench_default.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --amp > gpurun_out/r05/bench_amp.json 2> gpurun_out/r05/bench_amp.err
echo done
