cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" NB=1 N=61 MODES=graph python tools/scratch/graph_vs_eager.py 2>&1 | grep "^step" | awk '{printf "%s ", $4} END {print ""}' > gpurun_out/r41_$name.txt; }
run base X=1
run noprep DCD_CONV_PREP_TABLE=0
run noprepboth DCD_CONV_PREP_BOTH=0
run nomoments DCD_TRUNK_MOMENTS=0
run nowino DCD_CONV_WINOGRAD=0
run nodense DCD_DCN_DENSE=0
run nosweep DCD_BWD_SWEEP=0
run x3 DCD_PRECISION=bf16x3
