# Round-end measurement set (one GPU):  bash tools/final_measure.sh <tag>     -> gpurun_out/<tag>_*.json, step_<tag>*_kernels.csv
tag=$1
R=$GRAFT_REPO_ROOT
cd $R
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_f32.json 2> gpurun_out/${tag}_f32.err
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --amp > gpurun_out/${tag}_amp.json 2> /dev/null
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --precision bf16x3 > gpurun_out/${tag}_bf16x3.json 2> /dev/null
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --batch 1 > gpurun_out/${tag}_b1.json 2> /dev/null
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --batch 1 --amp > gpurun_out/${tag}_b1_amp.json 2> /dev/null
DCD_FORCE_DDP=1 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --batch 1 > gpurun_out/${tag}_ddp_b1.json 2> /dev/null
DCD_FORCE_DDP=1 python bench.py --steps 10 --warmup 4 --no-cpu-baseline > gpurun_out/${tag}_ddp_b8.json 2> /dev/null
DCD_FORCE_DDP=1 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --amp > gpurun_out/${tag}_ddp_b8_amp.json 2> /dev/null
python bench.py --workload gmw --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_gmw.json 2> /dev/null
python bench.py --workload gen --batch 16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_gen.json 2> /dev/null
bash tools/prof_step.sh ${tag} > gpurun_out/${tag}_prof.log 2>&1
bash tools/prof_step.sh ${tag}_b1 --batch 1 > gpurun_out/${tag}_prof_b1.log 2>&1
bash tools/prof_step.sh ${tag}_amp --amp > gpurun_out/${tag}_prof_amp.log 2>&1
cd $R
PMC_STEPS=4 python3 tools/pmc_kernels.py gpurun_out/${tag}_dcn_pmc.json "dcn_|128, false>" -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-split-line --no-op-line > gpurun_out/${tag}_pmc.log 2>&1
PMC_STEPS=4 python3 tools/pmc_kernels.py gpurun_out/${tag}_wino_pmc.json "wino_" -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-split-line --no-op-line > gpurun_out/${tag}_wino_pmc.log 2>&1
PMC_STEPS=4 python3 tools/pmc_kernels.py gpurun_out/${tag}_amp_pmc.json "dcn_|wino_|sgemm_bf16" -- python3 bench.py --steps 2 --warmup 2 --no-cpu-baseline --amp > gpurun_out/${tag}_amp_pmc.log 2>&1
# BASELINE.md section 3 also names bs 8 for the CPU baseline: opt-in (about 5 minutes of host time)
if [ "$2" = "cpu8" ]; then
  python bench.py --steps 5 --warmup 3 --no-split-line --no-op-line --cpu-batch 8 --cpu-budget 120 --cpu-timeout 900 > gpurun_out/${tag}_f32_cpu8.json 2> /dev/null
fi
for f in f32 amp bf16x3 b1 b1_amp ddp_b1 ddp_b8 ddp_b8_amp gmw gen; do python3 - <<P
import json
try:
    d=json.load(open("gpurun_out/${tag}_$f.json"))
    r=d.get("roofline",{})
    print("$f", round(d["value"],2), d["unit"], round(d["ms_per_step"],2), "ms | roofline", r.get("ms_per_step"), r.get("frac"))
except Exception as e:
    print("$f", "failed", e)
P
done
# round 6: the probes DESIGN.md section R6 cites
python tools/time_conv1x1.py 8 2>&1 | grep -v amdgpu > gpurun_out/${tag}_conv1x1_f32.txt
bash tools/route_probe.sh 2>&1 | grep -v amdgpu > gpurun_out/${tag}_route_probe.txt
python tools/offset_std_probe.py 0.01 0.033 0.06 0.08 0.1 2>&1 | grep "^std" > gpurun_out/${tag}_offset_std.txt
