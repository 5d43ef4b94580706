# rocprofv3 kernel trace of one DCN layer's forward+backward (round 3: one-pass backward); CSVs -> gpurun_out/r03_layer_*.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "64 64 96 320" "128 64 48 160"; do
  tag=$(echo $cfg | tr ' ' '_')
  rm -rf /tmp/prof_${tag}
  rocprofv3 --kernel-trace -d /tmp/prof_${tag} -- python3 $R/tools/one_layer.py $cfg > /dev/null 2>&1
  python3 $R/tools/prof_summary.py $(dirname $(find /tmp/prof_${tag} -name "*.db" | head -1)) $R/gpurun_out/r03_layer_${tag}${SUFFIX}.csv
done
