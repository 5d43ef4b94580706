cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_heads.py -x -q -m gpu -k edge_branch 2>&1 | grep -E "Error|assert|what|error" | head -20 > gpurun_out/r11_tests.txt
