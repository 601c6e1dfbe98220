#!/bin/bash
# A/B of the phase split of the phore-phore list launch (DP_PP_PSPLIT_MAX_CHUNKS) on one box: N samples of one complex
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/psplit_ab.txt; : > $out
for rep in 1 2; do
  for lib in "" build_abl/libnopsplit.so; do
    for cid in 2 1; do
      for P in 1 8 20 40 64 96 128 200; do
        v=$(DIFFPHORE_LIB=$lib python tools/graph_timeline_run.py $P $cid 2>/dev/null | tail -1)
        echo "lib=${lib:-product} $v" >> $out
      done
    done
  done
done
cat $out
