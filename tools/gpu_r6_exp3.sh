#!/bin/bash
# round 6, experiment 3: round-5 library (built from the round-5 commit) vs the product build vs DP_S0_TIMING (normals convs = their own s0-only launch, no listed pairs)
export PYTHONPATH=$PWD DIFFPHORE_ALLOW_ABLATION=1
mkdir -p gpurun_out
B="python bench.py --steps 60 --warmup 5 --no-alt --no-e2e --no-other --no-cpu-baseline"
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), d['roofline']['kernel_ms'], [x['ms'] for x in d['roofline']['step_weighted']['launches']])" $1; }
for rep in 1 2; do
for n in "$@"; do
  if [ $n == product ]; then $B > gpurun_out/exp3_$n.json 2> gpurun_out/exp3_$n.err; else DIFFPHORE_LIB=build_abl/lib$n.so $B > gpurun_out/exp3_$n.json 2> gpurun_out/exp3_$n.err; fi
  echo "$n $(val gpurun_out/exp3_$n.json)"
done
done | tee gpurun_out/exp3.txt
