#!/bin/bash
# A/B of the third-stream threshold for the phore-phore conv (DP_PP_BESIDE_MAX_WG) on one box: N samples of one complex, cfg2 and cfg1 shapes
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/ppbeside_ab.txt; : > $out
for rep in 1 2; do
  for lib in "" build_abl/libppb1200.so build_abl/libppb2400.so; do
    for cid in 2 1; do
      for P in 20 40 64 96; do
        v=$(DIFFPHORE_LIB=$lib python tools/graph_timeline_run.py $P $cid 2>/dev/null | tail -1)
        echo "lib=${lib:-product} $v" >> $out
      done
    done
  done
done
cat $out
