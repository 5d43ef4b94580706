#!/bin/bash
mkdir -p gpurun_out/r05
cd $GRAFT_REPO_ROOT
timeout 600 python tools/scratch/time_stride2.py > gpurun_out/r05/stride2_v2.txt 2>&1
DCD_CONV_S2D=1 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-line > gpurun_out/r05/bench4_s2d.json 2> gpurun_out/r05/bench4_s2d.err
DCD_CONV_S2D=0 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-line > gpurun_out/r05/bench4_nos2d.json 2> gpurun_out/r05/bench4_nos2d.err
(timeout 900 python -m pytest tests/test_gpu_golden.py -x -q -k "fp16" 2>&1 | grep -E "^E |passed|failed|Error" | head -30) > gpurun_out/r05/t4_fp16.txt
echo done
