#!/bin/bash
# small batches: ms per step with the normals convs split (DIFFPHORE_XN_SPARSE=1) and dense (0), node-side columns of tools/batch_sweep.py
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_c_host_gpu.py tests/test_forward_gpu.py -q -m gpu -k "range or replay" 2>&1 | tail -3
for xn in 0 1; do
  DIFFPHORE_XN_SPARSE=$xn python tools/batch_sweep.py --batches ${1:-1,8,20,40,64,96,128,192,256} --reps 3 --out gpurun_out/sweep_xn$xn.csv > /dev/null 2>&1
  echo "xn_sparse=$xn"; cut -d, -f1,6-9 gpurun_out/sweep_xn$xn.csv
done
