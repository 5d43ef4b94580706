cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" COMPACT=1 N=41 python tools/scratch/graph_twin.py 2>&1 | grep "^step" | awk '{printf "%s(%s) ", $4, $NF} END {print ""}' > gpurun_out/r49_$name.txt; }
run base X=1
run noedge DCD_EDGE_BRANCH_GEMM=0
run noheadrows DCD_HEAD_ROWS=0
run nostatk DCD_TRUNK_STATS_KERNELS=0
run nopatch DCD_TRUNK_PATCH_NODE=0
run grambmm DCD_TRUNK_GRAM=bmm
run noheadfused DCD_HEAD_FUSED=0
run nolossrows DCD_LOSS_ROWS=0
