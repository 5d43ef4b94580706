cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_conv.py -x -q -k "conv1x1" 2>&1 | tail -15
python tools/time_conv1x1.py 8 2>&1 | grep -v amdgpu.ids
python tools/time_conv1x1.py 1 2>&1 | grep -v amdgpu.ids
