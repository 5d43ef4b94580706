cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "64 64 96 320" "128 64 48 160" "256 256 24 80"; do
  tag=$(echo $cfg | tr ' ' '_')
  for mode in f32 bf16x3 gen1; do
    if [ $mode = gen1 ]; then export DCD_DW_GEN=1 DCD_PREC=f32; else export DCD_DW_GEN=2 DCD_PREC=$mode; fi
    rm -rf /tmp/prof_${tag}_${mode}
    rocprofv3 --kernel-trace -d /tmp/prof_${tag}_$mode -- python3 $R/tools/one_layer.py $cfg > /dev/null 2>&1
    python3 $R/tools/prof_summary.py $(dirname $(find /tmp/prof_${tag}_$mode -name "*.db" | head -1)) $R/gpurun_out/r02_layer_${tag}_$mode.csv
  done
done
