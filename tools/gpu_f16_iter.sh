#!/bin/bash
# one optimisation iteration of the f16x3 node-side kernel: parity, then the conv micro-benchmark (layer kinds 3 and 2, both cross modes)
mkdir -p gpurun_out
python -m pytest tests/test_conv_f16_gpu.py -q -x 2>&1 | tail -3
export PYTHONPATH=$PWD
for kind in 3 2 1 0; do
  for mode in p2l l2p; do
    python tools/bench_conv.py --algo node --prec f16x3 --kind $kind --mode $mode --iters 50 2>&1 | tail -1 | python -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['kind'], d['mode'], round(d['ms'], 4))"
  done
done
