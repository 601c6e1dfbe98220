#!/bin/bash
# evidence of the tree after the edge-side translation-unit split / StageP / exact replay (the node-side kernels are untouched: their PMC record,
# batch sweeps and the configs[2] / configs[3] traces of tools/gpu_round5_final3.sh stand): GPU suite record, kernel trace + HBM traffic of the
# default and the configs[4] bench (traffic records merged ON THE BOX), then the bench line that carries them
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
( python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > gpurun_out/r5_gpu_suite.txt
bash tools/profile_round.sh r5 > gpurun_out/r5_profile.log 2>&1 < /dev/null
bash tools/profile_round.sh r5_cfg5_bf16 --config 5 > gpurun_out/r5_cfg5.log 2>&1 < /dev/null
python tools/merge_traffic_record.py gpurun_out/r5_traffic_record.json gpurun_out/r5_cfg5_bf16_traffic_record.json
cp profiles/conv_hbm_traffic.json gpurun_out/conv_hbm_traffic.json
python bench.py > gpurun_out/r5_bench_full.json 2> gpurun_out/r5_bench_full.err < /dev/null
tail -3 gpurun_out/r5_gpu_suite.txt; tail -c 300 gpurun_out/r5_bench_full.json
