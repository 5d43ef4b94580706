cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/stats_r05
rocprofv3 --kernel-trace --stats -d /tmp/stats_r05 --output-format csv -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-split-line --no-op-line > $R/gpurun_out/r35_bench.json 2> /dev/null
f=$(find /tmp/stats_r05 -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/r35_kernel_stats.csv
ls -la /tmp/stats_r05/* | head > $R/gpurun_out/r35_ls.txt
