#!/bin/bash
# round-5 evidence in one call, in the order the numbers depend on each other: GPU suite record; rocprofv3 kernel trace + HBM traffic of the
# bench (-> the traffic record, merged into profiles/conv_hbm_traffic.json ON THE BOX so that the bench line that follows carries it);
# the default bench line; per-config traces; PMC of the f16x3 node-side kernel; the growth tables of the 100-step test.
# Everything lands in gpurun_out/ (copied to profiles/ by hand, conv_hbm_traffic.json included).
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
( python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > gpurun_out/r5_gpu_suite.txt
bash tools/profile_round.sh r5 > gpurun_out/r5_profile.log 2>&1
bash tools/profile_round.sh r5_cfg5_bf16 --config 5 > gpurun_out/r5_cfg5.log 2>&1
python tools/merge_traffic_record.py gpurun_out/r5_traffic_record.json $(ls gpurun_out/r5_cfg5_bf16_traffic_record.json 2>/dev/null)
cp profiles/conv_hbm_traffic.json gpurun_out/conv_hbm_traffic.json
python bench.py > gpurun_out/r5_bench_full.json 2> gpurun_out/r5_bench_full.err
python -m pytest tests/test_workloads_gpu.py -q -s -k "full_100 or matches_oracle" 2>&1 | grep -E "configs\[4\]|after|config [0-9]|passed|failed" > gpurun_out/r5_configs4_100step.txt
bash tools/profile_config.sh r5_cfg3 3 > gpurun_out/r5_cfg3.log 2>&1
bash tools/profile_config.sh r5_cfg4 4 > gpurun_out/r5_cfg4.log 2>&1
bash tools/pmc_conv.sh r5f16k3 --algo node --prec f16x3 --kind 3 --mode p2l > /dev/null 2>&1
python tools/batch_sweep.py --batches 1,8,20,40,64,96,128,256,512 --reps 2 --out gpurun_out/r5_batch_sweep.csv > /dev/null 2>&1 < /dev/null
python tools/batch_sweep.py --precision bf16 --batches 1,8,20,40,64,128,256,512 --reps 2 --out gpurun_out/r5_batch_sweep_bf16.csv > /dev/null 2>&1 < /dev/null
bash tools/pmc_bench.sh r5lig3 "tp_conv_node_f16_kernel<3, true" > /dev/null 2>&1
tail -3 gpurun_out/r5_gpu_suite.txt; tail -c 400 gpurun_out/r5_bench_full.json; ls gpurun_out | grep "^r5\|pmc_r5\|pmcb_r5"
