# usage: prof_script.sh <tag> <script.py> [args]   -> gpurun_out/<tag>_kernels.csv: per-kernel summary (rocprofv3 --kernel-trace) of the whole script
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace -d /tmp/prof_$tag -- python3 $R/"$@" > /tmp/prof_$tag.log 2>&1
python3 $R/tools/prof_summary.py $(dirname $(find /tmp/prof_$tag -name "*.db" | head -1)) $R/gpurun_out/${tag}_kernels.csv
