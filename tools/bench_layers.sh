#!/bin/bash
# bench.py with the per-layer DCN table:  tools/bench_layers.sh <tag> [extra bench args]
tag=$1; shift
python bench.py --steps 10 --warmup 4 --cpu-timeout 1 "$@" > gpurun_out/r02_bench_$tag.json 2>gpurun_out/r02_bench_$tag.err
python - <<P
import json
d=json.load(open("gpurun_out/r02_bench_$tag.json"))
print("step %.2f ms  %.1f img/s  dcn %.2f ms frac %.3f" % (d["ms_per_step"], d["value"], d["roofline"]["ms_per_step"], d["roofline"]["frac"]))
for k,v in d["roofline"]["layers"].items(): print(" ", k, v)
P
