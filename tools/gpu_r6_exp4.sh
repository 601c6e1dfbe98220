#!/bin/bash
# round 6, experiment 4: the normals convs as s0-only launch + listed pairs - parity tests, then the step with / without (DIFFPHORE_XN_SPARSE)
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_forward_gpu.py tests/test_sample_loop_gpu.py -x -q -m gpu > gpurun_out/exp4_tests.txt 2>&1
tail -5 gpurun_out/exp4_tests.txt
B="python bench.py --steps 60 --warmup 5 --no-alt --no-e2e --no-other --no-cpu-baseline"
val() { python -c "import json,sys; d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4), d['roofline']['kernel_ms'], [x['ms'] for x in d['roofline']['step_weighted']['launches']])" $1; }
for rep in 1 2; do
for n in 0 auto; do
  DIFFPHORE_XN_SPARSE=$n $B > gpurun_out/exp4_$n.json 2> gpurun_out/exp4_$n.err
  echo "xn_sparse=$n $(val gpurun_out/exp4_$n.json)"
done
done | tee gpurun_out/exp4.txt
