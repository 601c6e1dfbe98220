#!/bin/bash
# small batches: third stream on / off (DIFFPHORE_THIRD_STREAM), parity of the small-batch tests, timeline at 40 poses
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_forward_gpu.py tests/test_sample_loop_gpu.py -x -q -m gpu 2>&1 | tail -3
for t3 in 0 1; do
  DIFFPHORE_THIRD_STREAM=$t3 python tools/batch_sweep.py --batches ${1:-1,8,20,40,64,96} --reps 3 --out gpurun_out/sweep_t3_$t3.csv > /dev/null 2>&1
  echo "third_stream=$t3"; cut -d, -f1,6-9 gpurun_out/sweep_t3_$t3.csv
done
