cd $GRAFT_REPO_ROOT
DCD_STEP_GRAPH=1 bash tools/prof_step.sh g_amp --amp > /dev/null 2>&1
DCD_FORCE_DDP=1 bash tools/prof_step.sh g_ddp > /dev/null 2>&1
bash tools/prof_step.sh g_b1 --batch 1 > /dev/null 2>&1
DCD_FORCE_DDP=1 bash tools/prof_step.sh g_ddp_b1 --batch 1 > /dev/null 2>&1
DCD_STEP_GRAPH=1 bash tools/prof_step.sh g_x3 --precision bf16x3 > /dev/null 2>&1
for t in g_amp g_ddp g_b1 g_ddp_b1 g_x3; do echo "$t memsets: $(grep -c fillBuffer gpurun_out/r05_step_${t}_sequence.txt)  $(python -c "import json;d=json.load(open('gpurun_out/r05_step_${t}.json'));print(d['config']['step_launch'], round(d['ms_per_step'],2))")"; done > gpurun_out/r68_memsets.txt
for t in g_amp g_ddp g_b1 g_ddp_b1 g_x3; do grep -B3 -A2 fillBuffer gpurun_out/r05_step_${t}_sequence.txt | cut -c1-100 > gpurun_out/r68_ctx_$t.txt; done
