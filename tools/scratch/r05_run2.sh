#!/bin/bash
# round 5, GPU run 2: bf16 weight-gradient kernel without spills; per-kernel profile of the fp32 and the bf16 step
mkdir -p gpurun_out/r05
cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_dcn.py tests/test_gpu_conv.py -x -q -k "bf16 or policy" 2>&1 | tail -8) > gpurun_out/r05/t2_ops.txt
(timeout 900 python -m pytest tests/test_gpu_golden.py -x -q -k "fp16 or mixed_bf16 or baseline_size" 2>&1 | tail -15) > gpurun_out/r05/t2_model.txt
DCD_PRECISION=bf16 timeout 600 python tools/time_conv.py > gpurun_out/r05/conv_bf16_v2.txt 2>&1
bash tools/prof_step.sh f32 > gpurun_out/r05/prof_f32.log 2>&1
bash tools/prof_step.sh bf16 --precision bf16 > gpurun_out/r05/prof_bf16.log 2>&1
echo done
