# per-kernel times of one DCN layer, dense path on/off:  bash tools/prof_dense.sh "256 256 24 80" "128 128 48 160" ...   (MODES="1 0", OSC=0.5)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  tag=$(echo $cfg | tr ' ' '_')
  for mode in ${MODES:-1 0}; do
    export DCD_DCN_DENSE=$mode
    rm -rf /tmp/prof_${tag}_d$mode
    rocprofv3 --kernel-trace -d /tmp/prof_${tag}_d$mode -- python3 $R/tools/one_layer.py $cfg 8 ${OSC:-0.5} > /dev/null 2>&1
    python3 $R/tools/prof_summary.py $(dirname $(find /tmp/prof_${tag}_d$mode -name "*.db" | head -1)) $R/gpurun_out/r02_dense_${tag}_d$mode.csv > /dev/null
  done
done
