cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_dcn.py -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3 > gpurun_out/r16_tests.txt
for b in 1 2 4 8; do python tools/time_dcn_layers.py $b f32 0.5 2>/dev/null | grep -v amdgpu > gpurun_out/r16_b$b.txt; DCD_SWEEP_MIN_ROWS=8 python tools/time_dcn_layers.py $b f32 0.5 2>/dev/null | grep TOTAL > gpurun_out/r16_b${b}_old.txt; done
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-split-line --no-op-line --batch 1 > gpurun_out/r16_bench_b1.json 2>/dev/null
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-split-line --no-op-line > gpurun_out/r16_bench_b8.json 2>/dev/null
