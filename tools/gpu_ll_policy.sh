#!/bin/bash
# ligand-ligand conv: node-side (pair-dense passes inside the fused ligand-side launch) vs edge-side (two-limb fp16 kernel), per workload
export PYTHONPATH=$PWD
for cfg in 3 4 2 1; do
  echo "== bench.py --config $cfg"
  BENCH_ARGS="--config $cfg" bash tools/gpu_ab_env.sh - "DIFFPHORE_LL_NODE_MAX_ATOMS=24" "DIFFPHORE_LL_NODE_MAX_ATOMS=32" "DIFFPHORE_LL_NODE_MAX_ATOMS=64"
done
