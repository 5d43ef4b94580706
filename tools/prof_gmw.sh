# kernel summary of the GMW step:  bash tools/prof_gmw.sh <tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1
rm -rf /tmp/profg_$tag
rocprofv3 --kernel-trace -d /tmp/profg_$tag -- python3 $R/bench.py --workload gmw --steps 4 --warmup 3 --no-cpu-baseline > $R/gpurun_out/gmw_${tag}.json 2> /dev/null
python3 $R/tools/prof_summary.py $(dirname $(find /tmp/profg_$tag -name "*.db" | head -1)) $R/gpurun_out/gmw_${tag}_kernels.csv 3
