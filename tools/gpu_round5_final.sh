#!/bin/bash
# round-5 evidence in one call: GPU suite record, default bench line, rocprofv3 kernel trace + HBM traffic of the bench, per-config traces, PMC of the
# f16x3 node-side kernel, the printed growth tables of the 100-step test.  Everything lands in gpurun_out/ (copied to profiles/ by hand).
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
( python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > gpurun_out/r5_gpu_suite.txt
python -m pytest tests/test_workloads_gpu.py -q -s -k "full_100 or matches_oracle" 2>&1 | grep -E "configs\[4\]|after|config [0-9]|passed|failed" > gpurun_out/r5_configs4_100step.txt
python bench.py > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err
bash tools/profile_round.sh r5 > gpurun_out/r5_profile.log 2>&1
bash tools/profile_config.sh r5_cfg3 3 > gpurun_out/r5_cfg3.log 2>&1
bash tools/profile_config.sh r5_cfg4 4 > gpurun_out/r5_cfg4.log 2>&1
bash tools/profile_config.sh r5_cfg5_bf16 5 > gpurun_out/r5_cfg5.log 2>&1
bash tools/pmc_conv.sh r5f16k3 --algo node --prec f16x3 --kind 3 --mode p2l > /dev/null 2>&1
bash tools/pmc_conv.sh r5f32k3 --algo node --prec f32 --kind 3 --mode p2l > /dev/null 2>&1
bash tools/pmc_bench.sh r5lig3 "tp_conv_node_f16_kernel<3, true" > /dev/null 2>&1
tail -3 gpurun_out/r5_gpu_suite.txt; tail -c 300 gpurun_out/r5_bench.json; ls gpurun_out | grep "^r5\|pmc_r5\|pmcb_r5"
