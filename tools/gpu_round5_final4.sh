#!/bin/bash
# the bf16 workload's PMC traffic record + the bench line that carries it (after tools/gpu_round5_final3.sh, whose records are already in profiles/)
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/profile_round.sh r5_cfg5_bf16 --config 5 > gpurun_out/r5_cfg5.log 2>&1
python tools/merge_traffic_record.py gpurun_out/r5_cfg5_bf16_traffic_record.json
cp profiles/conv_hbm_traffic.json gpurun_out/conv_hbm_traffic.json
python bench.py > gpurun_out/r5_bench_full.json 2> gpurun_out/r5_bench_full.err
tail -c 300 gpurun_out/r5_bench_full.json; cat gpurun_out/r5_cfg5_bf16_traffic_record.json
