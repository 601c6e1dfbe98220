#!/bin/bash
# first GPU contact of the f16x3 node-side kernel: parity tests, then the conv micro-benchmark against the fp32 and bf16 node-side kernels
mkdir -p gpurun_out
python -m pytest tests/test_conv_f16_gpu.py -q -s -x 2>&1 | tail -60 > gpurun_out/f16_conv_test.log
tail -8 gpurun_out/f16_conv_test.log
export PYTHONPATH=$PWD
for kind in 3 2 1 0; do
  for prec in f32 bf16 f16x3; do
    for mode in p2l l2p; do
      python tools/bench_conv.py --algo node --prec $prec --kind $kind --mode $mode 2>&1 | tail -1
    done
  done
done > gpurun_out/f16_conv_bench.log 2>&1
cat gpurun_out/f16_conv_bench.log
