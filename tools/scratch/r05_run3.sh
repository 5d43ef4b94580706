#!/bin/bash
mkdir -p gpurun_out/r05
cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_golden.py -x -q -k "fp16" 2>&1 | grep -E "^E |passed|failed|Error" | head -30) > gpurun_out/r05/t3_fp16.txt
timeout 600 python tools/scratch/time_stride2.py > gpurun_out/r05/stride2.txt 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-line > gpurun_out/r05/bench3_default.json 2> gpurun_out/r05/bench3_default.err
timeout 900 python tools/time_dcn_layers.py 8 f32 0.5 > gpurun_out/r05/dcn_layers_f32.txt 2>&1; timeout 900 python tools/time_dcn_layers.py 8 bf16 0.5 > gpurun_out/r05/dcn_layers_bf16.txt 2>&1
echo done
