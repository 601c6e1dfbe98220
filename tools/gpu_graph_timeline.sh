#!/bin/bash
# timeline of one step of the N-samples-of-one-complex call in its shipped driver (hipGraph replay + second stream): tools/gpu_graph_timeline.sh [poses] [config id]
P=${1:-40}; CID=${2:-2}
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out; rm -rf /tmp/prof_g; mkdir -p /tmp/prof_g
R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d /tmp/prof_g -o t -f csv -- python $R/tools/graph_timeline_run.py $P $CID > /tmp/prof_g/out.txt 2>&1)
T=$(find /tmp/prof_g -name '*kernel_trace.csv' | head -1)
python tools/step_timeline.py "$T" --step 90 > gpurun_out/graph${P}_cfg${CID}_timeline.txt
cat gpurun_out/graph${P}_cfg${CID}_timeline.txt; tail -2 /tmp/prof_g/out.txt
python tools/graph_timeline_run.py $P $CID | tail -1
