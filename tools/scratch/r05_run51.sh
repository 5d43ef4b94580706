cd $GRAFT_REPO_ROOT
run() { name=$1; shift; env "$@" python tools/scratch/graph_perturb.py 2>&1 | grep -E "median" > gpurun_out/r51_$name.txt; }
run base X=1
run noheadrows DCD_HEAD_ROWS=0
run nopatch DCD_TRUNK_PATCH_NODE=0
run nolossrows DCD_LOSS_ROWS=0
run noedge DCD_EDGE_BRANCH_GEMM=0
run nostatk DCD_TRUNK_STATS_KERNELS=0
run noheadfused DCD_HEAD_FUSED=0
run eps0 EPS=0
