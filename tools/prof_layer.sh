# usage: prof_layer.sh <tag> <C Co H W [B] [off_scale]> ; env passes through.  Per-kernel CSV of one DCN layer (fwd+bwd x4) ->
# gpurun_out/r04_layer_<tag>.csv   (kernel trace only: no counters in this pass)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace -d /tmp/prof_$tag -- python3 $R/tools/one_layer.py "$@" > /dev/null 2>&1
python3 $R/tools/prof_summary.py $(dirname $(find /tmp/prof_$tag -name "*.db" | head -1)) $R/gpurun_out/r04_layer_$tag.csv
