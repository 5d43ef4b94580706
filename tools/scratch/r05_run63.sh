cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_golden.py -q -m gpu -k "eager_twin" 2>&1 | grep -E "passed|failed|Error|assert" | head -5 > gpurun_out/r63_tests.txt
git stash -q 2>/dev/null
