#!/bin/bash
# The round's evidence in ONE gpurun call, in the order the numbers depend on each other (usage: tools/gpu_evidence.sh <tag, e.g. r6>):
#   1. the GPU suite record;
#   2. rocprofv3 kernel trace + stats and the two HBM-traffic PMC passes of the bench step (tools/profile_round.sh) -> the traffic records, merged into
#      profiles/conv_hbm_traffic.json ON THE BOX so that the bench line that follows quotes them (they are keyed to the kernel sources' hash);
#   3. the default bench line (what the driver runs);
#   4. per-config kernel traces, the small-batch sweeps, PMC of the dominant launches, the 100-step growth table, the single-rank RCCL check.
# Everything lands in gpurun_out/<tag>_*; copy what is to be judged into profiles/ (conv_hbm_traffic.json included) and commit.
tag=${1:-r6}
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
( python -m pytest tests -m gpu -q 2>&1 | tail -5 ) > gpurun_out/${tag}_gpu_suite.txt
bash tools/profile_round.sh ${tag} > gpurun_out/${tag}_profile.log 2>&1
bash tools/profile_round.sh ${tag}_cfg5_bf16 --config 5 > gpurun_out/${tag}_cfg5.log 2>&1
python tools/merge_traffic_record.py gpurun_out/${tag}_traffic_record.json $(ls gpurun_out/${tag}_cfg5_bf16_traffic_record.json 2>/dev/null)
cp profiles/conv_hbm_traffic.json gpurun_out/conv_hbm_traffic.json
python bench.py > gpurun_out/${tag}_bench_full.json 2> gpurun_out/${tag}_bench_full.err
python -m pytest tests/test_workloads_gpu.py -q -s -k "full_100 or matches_oracle" 2>&1 | grep -E "configs\[4\]|after|config [0-9]|passed|failed" > gpurun_out/${tag}_configs4_100step.txt
bash tools/profile_config.sh ${tag}_cfg1 1 > gpurun_out/${tag}_cfg1.log 2>&1
bash tools/profile_config.sh ${tag}_cfg3 3 > gpurun_out/${tag}_cfg3.log 2>&1
bash tools/profile_config.sh ${tag}_cfg4 4 > gpurun_out/${tag}_cfg4.log 2>&1
python tools/batch_sweep.py --batches 1,8,20,40,64,96,128,256,512 --reps 2 --out gpurun_out/${tag}_batch_sweep.csv > /dev/null 2>&1 < /dev/null
python tools/batch_sweep.py --precision bf16 --batches 1,8,20,40,64,128,256,512 --reps 2 --out gpurun_out/${tag}_batch_sweep_bf16.csv > /dev/null 2>&1 < /dev/null
bash tools/pmc_bench.sh ${tag}lig3 "tp_conv_node_f16_kernel<3, true" > /dev/null 2>&1
bash tools/pmc_bench.sh ${tag}s0lig3 "tp_conv_node_f16_kernel<3, false, 1, false, true" > /dev/null 2>&1
bash tools/pmc_conv.sh ${tag}f16k3 --algo node --prec f16x3 --kind 3 --mode p2l > /dev/null 2>&1
bash tools/gpu_step_timeline.sh 40 > gpurun_out/${tag}_small40.log 2>&1
# N samples of ONE complex (the reference's own call): timeline of a step in the captured-graph driver, eager / captured / replayed loop times, shared-phore A/B
for pc in "40 2" "1 2" "40 1"; do bash tools/gpu_graph_timeline.sh $pc > /dev/null 2>&1; done
python tools/graph_create_cost.py 1 8 40 128 512 2>/dev/null | grep poses > gpurun_out/${tag}_loop_drivers.txt
bash tools/gpu_phshare_ab.sh > /dev/null 2>&1; cp gpurun_out/phshare_ab.txt gpurun_out/${tag}_phshare_ab.txt
python tools/rccl_single_rank_check.py > gpurun_out/${tag}_rccl_single_rank.txt 2>&1
tail -3 gpurun_out/${tag}_gpu_suite.txt; tail -c 400 gpurun_out/${tag}_bench_full.json; ls gpurun_out | grep "^${tag}\|pmc_${tag}\|pmcb_${tag}"
