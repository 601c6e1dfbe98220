#!/bin/bash
# A/B of library variants on the headline workload: tools/gpu_ab_bench.sh <variant lib or "base"> ...   (bench.py --steps 60, value + ms/step)
export PYTHONPATH=$PWD
for v in "$@"; do
  if [ "$v" == "base" ]; then unset DIFFPHORE_LIB; else export DIFFPHORE_LIB=$v; fi
  python bench.py --steps 60 --warmup 5 --no-alt --no-e2e --no-cpu-baseline --no-other 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value']), round(d['ms_per_step'], 4))"
done
