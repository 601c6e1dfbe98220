#!/bin/bash
# Same-box A/B of the product library against variant builds (tools/build_variant.sh) on the small-batch loop, alternating arms, two repetitions:
#   tools/gpu_variant_ab.sh <tool> "<poses ...>" <config id> [lib.so ...]        ("" as a lib = the product)
#   tool = timeline : tools/graph_timeline_run.py (N samples of one complex, graph replay + second stream, ms per step)
#          drivers  : tools/graph_create_cost.py (eager / captured / replayed 20-step loop, ms)
# This round's thresholds came from it: DP_PP_BESIDE_MAX_WG (-DDP_PP_BESIDE_MAX_WG=1200 / 2400), DP_PP_PSPLIT_MAX_CHUNKS (=0), DP_HEADS_EARLY (=0 / 4096),
# DP_SYM_BARRIER (=0) - profiles/r6_experiments.md sections 7-8.  The 512-pose headline against variants: tools/gpu_lib_ab.sh.
tool=$1; poses=$2; cid=$3; shift 3
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/variant_ab.txt; : > $out
for rep in 1 2; do
  for lib in "" "$@"; do
    echo "lib=${lib:-product} rep=$rep" >> $out
    if [ "$tool" == "drivers" ]; then
      DIFFPHORE_LIB=$lib python tools/graph_create_cost.py $poses 2>/dev/null | grep "eager\|replay" >> $out
    else
      for P in $poses; do DIFFPHORE_LIB=$lib python tools/graph_timeline_run.py $P $cid 2>/dev/null | tail -1 >> $out; done
    fi
  done
done
cat $out
