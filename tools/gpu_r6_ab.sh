#!/bin/bash
# A/B of one environment switch on the default bench step: tools/gpu_r6_ab.sh VAR a b [bench args]
export PYTHONPATH=$PWD
mkdir -p gpurun_out
var=$1; va=$2; vb=$3; shift 3
B="python bench.py --steps 60 --warmup 5 --no-alt --no-e2e --no-other --no-cpu-baseline $*"
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), d['roofline']['kernel_ms'])" $1; }
for rep in 1 2 3; do for v in $va $vb; do
  env $var=$v $B > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  echo "$var=$v $(val gpurun_out/ab_$v.json)"
done; done | tee gpurun_out/ab_$var.txt
