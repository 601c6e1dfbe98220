#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests/test_workloads_gpu.py -q -s -k "full_100 or matches_oracle" 2>&1 | grep -E "configs\[4\]|after|config [0-9]|passed|failed" > gpurun_out/r5_configs4_100step.txt
python bench.py > gpurun_out/r5_bench_full.json 2> gpurun_out/r5_bench_full.err
bash tools/profile_round.sh r5 > gpurun_out/r5_profile.log 2>&1
tail -4 gpurun_out/r5_configs4_100step.txt; tail -c 300 gpurun_out/r5_bench_full.json; cat gpurun_out/r5_traffic_record.json; grep node_f16 gpurun_out/r5_conv_hbm_traffic.csv
