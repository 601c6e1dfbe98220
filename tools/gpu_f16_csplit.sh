#!/bin/bash
# small batches of the fp32 model: the f16x3 node-side kernel with conv split (one conv per workgroup + combine) against the round's default
# (fp32 path-split kernels up to 48 node groups, unsplit f16x3 launches beyond); variant libraries = thresholds in workgroups of the fused launch
export PYTHONPATH=$PWD
python -m pytest tests/test_forward_gpu.py tests/test_conv_f16_gpu.py -x -q 2>&1 | tail -2
for lib in "" cs256 cs512 cs1024; do
  echo "== ${lib:-product (threshold 0)}"
  if [ -n "$lib" ]; then export DIFFPHORE_LIB=$PWD/build_abl/lib$lib.so; else unset DIFFPHORE_LIB; fi
  python tools/batch_sweep.py --batches 1,8,20,40,64,96,128,192,256 --reps 2 --out gpurun_out/r5_f16_csplit_${lib:-off}.csv 2>&1 < /dev/null | grep poses | python -c "
import sys, ast
for l in sys.stdin:
    d = ast.literal_eval(l.strip()); print(d['poses'], 'node/loop', d['node/loop'], 'node/graph+2s', d['node/graph+2s'])"
done
