cd $GRAFT_REPO_ROOT
for i in 0 2 3 5; do SHAPE=$i bash tools/prof_script.sh pw_$i tools/time_conv1x1.py 8; grep -i "pw_\|Cijk\|reduce" gpurun_out/pw_${i}_kernels.csv | cut -c1-60,120- | head -12; done
