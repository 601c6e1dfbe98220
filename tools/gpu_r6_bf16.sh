#!/bin/bash
# the bf16 model with the normals convs split (s0-only bf16 launch + listed pairs): parity tests, then configs[4] with / without; small-batch sweep with the third stream's threshold
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_conv_bf16_gpu.py tests/test_workloads_gpu.py tests/test_forward_gpu.py tests/test_sample_loop_gpu.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --config 5 --steps 40 --warmup 5 --no-alt --no-e2e --no-other --no-cpu-baseline"
for rep in 1 2; do for n in 0 auto; do
  DIFFPHORE_XN_SPARSE=$n $B > gpurun_out/bf16_$n.json 2> gpurun_out/bf16_$n.err
  echo "config 5 xn_sparse=$n $(python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))" gpurun_out/bf16_$n.json)"
done; done
python tools/batch_sweep.py --batches 1,8,14,20,40 --reps 3 --out gpurun_out/sweep_final.csv > /dev/null 2>&1; cut -d, -f1,6-9 gpurun_out/sweep_final.csv
