#!/bin/bash
# small batches: the library default (fp32 path-split / conv-split kernels while B x groups <= 256) against the f16x3 node-side kernel forced on
# (DIFFPHORE_PATH_SPLIT=0); ms per step of the best driver per batch size
export PYTHONPATH=$PWD
for ps in auto 0; do
  echo "== DIFFPHORE_PATH_SPLIT=$ps"
  DIFFPHORE_PATH_SPLIT=$ps python tools/batch_sweep.py --batches 8,20,40,80,128,160,256 --reps 2 --out gpurun_out/r5_small_$ps.csv 2>&1 | tail -12
done
