cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_outl
rocprofv3 --kernel-trace -d /tmp/prof_outl -- python3 $R/tools/time_bwd_outliers.py > /dev/null 2>&1
python3 $R/tools/prof_summary.py $(dirname $(find /tmp/prof_outl -name "*.db" | head -1)) $R/gpurun_out/r02_outliers.csv > /dev/null
