#!/bin/bash
# kernel-trace + stats of one BASELINE config's workload: tools/profile_config.sh <tag> <bench --config value> [more bench args]
set -u
tag=$1; cfgv=$2; shift 2
cd "$(dirname "$0")/.." || exit 1
export PYTHONPATH=. TMPDIR=/tmp
ROOT=$PWD
mkdir -p gpurun_out
CMD="python bench.py --config $cfgv --steps 20 --warmup 3 --no-alt --no-e2e --no-cpu-baseline --no-other $*"
$CMD > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rm -rf /tmp/prof_$tag; mkdir -p /tmp/prof_$tag
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag/trace -o t -- python $ROOT/bench.py --config $cfgv --steps 20 --warmup 3 --no-alt --no-e2e --no-cpu-baseline --no-other $* > /tmp/prof_$tag/trace.out 2>&1)
T=$(find /tmp/prof_$tag/trace -name '*.db' | head -1)
python tools/summarize_prof.py gpurun_out/$tag "$T" --bench gpurun_out/${tag}_bench.json --command "rocprofv3 --kernel-trace --stats -- $CMD"
head -40 gpurun_out/${tag}_kernel_stats.csv
