#!/bin/bash
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_conv_f16_gpu.py tests/test_forward_gpu.py -x -q -m gpu -k "s0_only or list_launch or normals_convs" -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/t1.txt
cat gpurun_out/t1.txt
