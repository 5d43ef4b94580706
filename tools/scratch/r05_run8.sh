cd $GRAFT_REPO_ROOT
python tools/scratch/conv_bf16_probe.py > gpurun_out/r8_probe.txt 2>&1
DCD_PROBE_ONLY=1 python3 tools/pmc_kernels.py gpurun_out/r8_conv_pmc.json "wino_conv3x3_split|wino_wrw3x3" -- python3 tools/scratch/conv_bf16_probe.py > gpurun_out/r8_pmc.log 2>&1
