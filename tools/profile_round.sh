#!/bin/bash
# rocprofv3 passes on bench.py -> gpurun_out/<tag>_*: kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in their own runs.
# usage: tools/profile_round.sh <tag> [extra bench.py arguments, e.g. --config 4]      (default workload: configs[1])
set -u
tag=${1:-r2}
shift || true
EXTRA="$*"
CID=2; prev=""; for w in $EXTRA; do [ "$prev" == "--config" ] && CID=$w; prev=$w; done
cd "$(dirname "$0")/.." || exit 1
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out
CMD="python bench.py --steps 40 --warmup 5 --no-alt --no-e2e --no-cpu-baseline --no-other $EXTRA"
$CMD > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
rm -rf /tmp/prof_$tag; mkdir -p /tmp/prof_$tag
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag/trace -o t -- python $OLDPWD/bench.py --steps 40 --warmup 5 --no-alt --no-e2e --no-cpu-baseline --no-other $EXTRA > /tmp/prof_$tag/trace.out 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/prof_$tag/fetch -o f -- python $OLDPWD/bench.py --steps 8 --warmup 2 --no-alt --no-e2e --no-cpu-baseline --no-other $EXTRA > /tmp/prof_$tag/fetch.out 2>&1)
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/prof_$tag/write -o w -- python $OLDPWD/bench.py --steps 8 --warmup 2 --no-alt --no-e2e --no-cpu-baseline --no-other $EXTRA > /tmp/prof_$tag/write.out 2>&1)
T=$(find /tmp/prof_$tag/trace -name '*.db' | head -1); F=$(find /tmp/prof_$tag/fetch -name '*.db' | head -1); W=$(find /tmp/prof_$tag/write -name '*.db' | head -1)
echo "dbs: $T $F $W"
python tools/summarize_prof.py gpurun_out/$tag "$T" --fetch "$F" --write "$W" --bench gpurun_out/${tag}_bench.json --config-id $CID --command "rocprofv3 --kernel-trace --stats -- $CMD (PMC passes: --steps 8 --warmup 2)"
tail -1 /tmp/prof_$tag/trace.out | cut -c1-300
# (append gpurun_out/${tag}_traffic_record.json to profiles/conv_hbm_traffic.json when the profile is committed)
head -30 gpurun_out/${tag}_conv_slots.csv; cat gpurun_out/${tag}_traffic_record.json; head -14 gpurun_out/${tag}_kernel_stats.csv
