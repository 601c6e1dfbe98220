#!/bin/bash
# timeline of one step of a small batch (default 40 poses): rocprofv3 kernel trace of bench.py --poses P
P=${1:-40}
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf /tmp/prof_small; mkdir -p /tmp/prof_small
R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_small -o t -f csv -- python $R/bench.py --poses $P --steps 40 --warmup 5 --no-alt --no-e2e --no-cpu-baseline --no-other > /tmp/prof_small/out.txt 2>&1)
T=$(find /tmp/prof_small -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py "$T" --step 20 > gpurun_out/small${P}_timeline.txt
cat gpurun_out/small${P}_timeline.txt
grep '^{' /tmp/prof_small/out.txt | python -c "import sys, json; d = json.loads(sys.stdin.read()); print('ms_per_step', d['ms_per_step'])"
