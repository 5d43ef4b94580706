#!/bin/bash
mkdir -p gpurun_out/r05
cd $GRAFT_REPO_ROOT
(timeout 3000 python -m pytest tests -m gpu -x -q 2>&1 | tail -25) > gpurun_out/r05/t5_all.txt
timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-line > gpurun_out/r05/bench5_default.json 2> gpurun_out/r05/bench5_default.err
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --amp > gpurun_out/r05/bench5_amp.json 2> gpurun_out/r05/bench5_amp.err
timeout 600 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --batch 1 > gpurun_out/r05/bench5_b1.json 2> gpurun_out/r05/bench5_b1.err
timeout 600 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --batch 1 --amp > gpurun_out/r05/bench5_b1_amp.json 2> gpurun_out/r05/bench5_b1_amp.err
bash tools/prof_step.sh bf16v2 --amp > gpurun_out/r05/prof_bf16v2.log 2>&1
echo done
