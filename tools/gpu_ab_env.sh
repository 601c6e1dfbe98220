#!/bin/bash
# A/B of environment settings on the headline workload: tools/gpu_ab_env.sh "VAR=val ..." "VAR=val ..." ...   ("-" = library defaults)
export PYTHONPATH=$PWD
for v in "$@"; do
  if [ "$v" == "-" ]; then e=""; else e="$v"; fi
  env $e python bench.py --steps 60 --warmup 5 --no-alt --no-e2e --no-cpu-baseline --no-other $BENCH_ARGS 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$v]', round(d['value']), round(d['ms_per_step'], 4))"
done
