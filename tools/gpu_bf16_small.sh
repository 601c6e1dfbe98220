#!/bin/bash
# small batches of the bf16 model: conv split of the node-side launches (one conv per workgroup + combine) on / off
export PYTHONPATH=$PWD
python -m pytest tests/test_conv_bf16_gpu.py tests/test_forward_gpu.py -x -q -k "bf16" 2>&1 | tail -3
for fuse in 1 nosplit; do
  echo "== DIFFPHORE_CONV_FUSE=$fuse"
  DIFFPHORE_CONV_FUSE=$fuse python tools/batch_sweep.py --precision bf16 --batches 1,8,20,40,64,128,256 --reps 2 --out gpurun_out/r5_small_bf16_$fuse.csv 2>&1 | tail -10
done
