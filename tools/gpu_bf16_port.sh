#!/bin/bash
export PYTHONPATH=$PWD
python -m pytest tests/test_conv_bf16_gpu.py tests/test_forward_gpu.py tests/test_rng_stream_gpu.py -x -q -k "bf16" 2>&1 | tail -3
for k in 3 2; do
python tools/bench_conv.py --algo node --prec bf16 --kind $k 2>&1 | tail -1 | cut -c1-120
python tools/bench_conv.py --algo node --prec bf16 --kind $k --no-tail 1 2>&1 | tail -1 | cut -c1-120
python tools/bench_conv.py --algo node --prec bf16 --kind $k --mode l2p --no-tail 1 2>&1 | tail -1 | cut -c1-120
done
BENCH_ARGS="--config 5" bash tools/gpu_ab_env.sh - -
BENCH_ARGS="--config 2 --precision bf16" bash tools/gpu_ab_env.sh - -
