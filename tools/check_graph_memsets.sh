# Memset nodes in the CAPTURED train step (on this stack they are not ordered reliably inside a replayed
# HIP graph).  Kernel trace of the graphed step in eight configurations; every count below must be 0.
#   bash tools/check_graph_memsets.sh            (on the GPU box; writes gpurun_out/graph_memsets.txt)
cd $GRAFT_REPO_ROOT
out=gpurun_out/graph_memsets.txt
: > $out
run() { tag=$1; shift; env "$@" bash tools/prof_step.sh $tag ${EXTRA} > /dev/null 2>&1; \
        echo "$tag: $(grep -c fillBuffer gpurun_out/step_${tag}_sequence.txt) memset launches per traced step ($(python3 -c "import json;d=json.load(open('gpurun_out/step_${tag}.json'));print(d['config']['step_launch'])"))" >> $out; }
EXTRA="" run memchk_b8 DCD_STEP_GRAPH=1
EXTRA="--amp" run memchk_amp DCD_STEP_GRAPH=1
EXTRA="--precision bf16x3" run memchk_x3 DCD_STEP_GRAPH=1
EXTRA="" run memchk_ddp DCD_FORCE_DDP=1
EXTRA="--batch 1" run memchk_b1 X=1
EXTRA="--batch 2" run memchk_b2 X=1
EXTRA="--batch 4" run memchk_b4 X=1
# a size whose deep maps are off the space-to-depth path's alignment rules (DLA level 5 at 6x20): zero-padded form
EXTRA="--batch 2 --input 320x96 --no-op-line --no-split-line" run memchk_96x320 X=1
cat $out
