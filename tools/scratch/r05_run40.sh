cd $GRAFT_REPO_ROOT
NB=1 N=80 MODES=graph python tools/scratch/graph_vs_eager.py 2>&1 | grep "^step" > gpurun_out/r40_base.txt
NB=1 N=80 MODES=graph RECAP=10 python tools/scratch/graph_vs_eager.py 2>&1 | grep "^step" > gpurun_out/r40_recap.txt
NB=1 N=80 MODES=graph DCD_EDGE_BRANCH_GEMM=0 python tools/scratch/graph_vs_eager.py 2>&1 | grep "^step" > gpurun_out/r40_noedge.txt
NB=1 N=80 MODES=graph DCD_DCN_HANDOVER=always python tools/scratch/graph_vs_eager.py 2>&1 | grep "^step" > gpurun_out/r40_always.txt
NB=1 N=80 MODES=eager python tools/scratch/graph_vs_eager.py 2>&1 | grep "^step" > gpurun_out/r40_eager.txt
