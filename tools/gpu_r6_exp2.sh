#!/bin/bash
# round 6, experiment 2: the s0-only conv behind a call - accumulators copied (product) vs passed in place (they then live in scratch everywhere)
export PYTHONPATH=$PWD DIFFPHORE_ALLOW_ABLATION=1
mkdir -p gpurun_out
B="python bench.py --steps 60 --warmup 5 --no-alt --no-e2e --no-other --no-cpu-baseline"
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), d['roofline']['kernel_ms'])" $1; }
for rep in 1 2; do
for n in product accin s0t s0tin; do
  if [ $n == product ]; then $B > gpurun_out/exp2_$n.json 2> gpurun_out/exp2_$n.err; else DIFFPHORE_LIB=build_abl/lib$n.so $B > gpurun_out/exp2_$n.json 2> gpurun_out/exp2_$n.err; fi
  echo "$n $(val gpurun_out/exp2_$n.json)"
done
done | tee gpurun_out/exp2.txt
