cd $GRAFT_REPO_ROOT
tag=r05c
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_f32.json 2> /dev/null
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --amp > gpurun_out/${tag}_amp.json 2> /dev/null
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --batch 1 > gpurun_out/${tag}_b1.json 2> /dev/null
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --batch 1 --amp > gpurun_out/${tag}_b1_amp.json 2> /dev/null
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --batch 2 > gpurun_out/${tag}_b2.json 2> /dev/null
python bench.py --steps 10 --warmup 4 --no-cpu-baseline --batch 4 > gpurun_out/${tag}_b4.json 2> /dev/null
DCD_FORCE_DDP=1 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --batch 1 > gpurun_out/${tag}_ddp_b1.json 2> /dev/null
DCD_FORCE_DDP=1 python bench.py --steps 10 --warmup 4 --no-cpu-baseline > gpurun_out/${tag}_ddp_b8.json 2> /dev/null
DCD_FORCE_DDP=1 python bench.py --steps 10 --warmup 4 --no-cpu-baseline --amp > gpurun_out/${tag}_ddp_b8_amp.json 2> /dev/null
