#!/bin/bash
# phore-phore convs as one list launch: parity, then the step, the other configs and the small-batch sweep
export PYTHONPATH=$PWD
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_conv_f16_gpu.py tests/test_forward_gpu.py tests/test_sample_loop_gpu.py tests/test_c_host_gpu.py -x -q -m gpu 2>&1 | tail -3
B="python bench.py --steps 60 --warmup 5 --no-alt --no-e2e --no-cpu-baseline"
$B > gpurun_out/pp_bench.json 2> gpurun_out/pp_bench.err
python - <<PY
import json
d = json.loads(open("gpurun_out/pp_bench.json").read().strip().splitlines()[-1])
print("value", round(d["value"]), round(d["ms_per_step"], 4), "other", [(o.get("config"), round(o.get("value", 0))) for o in d["config"]["other_configs"] or []])
PY
python tools/batch_sweep.py --batches 1,8,20,40,96 --reps 3 --out gpurun_out/sweep_pp.csv > /dev/null 2>&1; cut -d, -f1,6-9 gpurun_out/sweep_pp.csv
bash tools/gpu_r6_trace.sh pp | head -24
