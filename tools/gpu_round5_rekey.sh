#!/bin/bash
# the hash-keyed part of the evidence only (after a change that leaves the kernels' CODE as it was measured but not the key - e.g. the key's own
# definition): PMC traffic records of the fp32 and bf16 workloads, merged on the box, then the bench line that carries them
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
bash tools/profile_round.sh r5 > gpurun_out/r5_profile.log 2>&1 < /dev/null
bash tools/profile_round.sh r5_cfg5_bf16 --config 5 > gpurun_out/r5_cfg5.log 2>&1 < /dev/null
python tools/merge_traffic_record.py gpurun_out/r5_traffic_record.json gpurun_out/r5_cfg5_bf16_traffic_record.json
cp profiles/conv_hbm_traffic.json gpurun_out/conv_hbm_traffic.json
python bench.py > gpurun_out/r5_bench_full.json 2> gpurun_out/r5_bench_full.err < /dev/null
tail -c 300 gpurun_out/r5_bench_full.json
