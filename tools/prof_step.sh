# usage: prof_step.sh <tag> [bench args...]   kernel-trace of bench.py, summary of the last 3 steps -> gpurun_out/step_<tag>_kernels.csv
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
tag=$1; shift
rm -rf /tmp/profs_$tag
rocprofv3 --kernel-trace --memory-copy-trace -d /tmp/profs_$tag -- python3 $R/bench.py --steps 4 --warmup 3 --no-cpu-baseline --no-split-line --no-op-line "$@" > $R/gpurun_out/step_${tag}.json 2> $R/gpurun_out/step_${tag}.err
python3 $R/tools/prof_summary.py $(dirname $(find /tmp/profs_$tag -name "*.db" | head -1)) $R/gpurun_out/step_${tag}_kernels.csv 3
python3 $R/tools/prof_sequence.py $(dirname $(find /tmp/profs_$tag -name "*.db" | head -1)) $R/gpurun_out/step_${tag}_sequence.txt
python3 $R/tools/prof_gaps.py $(dirname $(find /tmp/profs_$tag -name "*.db" | head -1)) 1 20 > $R/gpurun_out/step_${tag}_gaps.txt 2>&1
