cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -5 > gpurun_out/r12_tests.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r12_smoke.txt 2>&1
python bench.py > gpurun_out/r12_bench_default.json 2> gpurun_out/r12_bench_default.err
