cd $GRAFT_REPO_ROOT
export MIOPEN_DEBUG_GROUP_CONV_IMPLICIT_GEMM_HIP_BWD_XDLOPS=0
DCD_STEP_GRAPH=1 bash tools/prof_step.sh graphb8c > gpurun_out/r62_prof.log 2>&1
grep -c fillBuffer gpurun_out/r05_step_graphb8c_sequence.txt > gpurun_out/r62_memsets.txt
grep -c "ck::" gpurun_out/r05_step_graphb8c_sequence.txt >> gpurun_out/r62_memsets.txt
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-split-line --no-op-line > gpurun_out/r62_f32.json 2>/dev/null
unset MIOPEN_DEBUG_GROUP_CONV_IMPLICIT_GEMM_HIP_BWD_XDLOPS
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --no-split-line --no-op-line > gpurun_out/r62_f32_default.json 2>/dev/null
