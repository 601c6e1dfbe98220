#!/bin/bash
# one stream vs dp_forward's second stream (phore-side convs, torsion head, ligand radius graph beside the ligand side), per workload
for cfg in 1 3 4 5; do
  echo "== bench.py --config $cfg"
  BENCH_ARGS="--config $cfg" bash tools/gpu_ab_env.sh - "DIFFPHORE_TWO_STREAMS=0" "DIFFPHORE_TWO_STREAMS=1"
done
