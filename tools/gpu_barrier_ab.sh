#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/barrier_ab.txt; : > $out
for rep in 1 2; do
  for lib in "" build_abl/libchained.so; do
    echo "lib=${lib:-product} rep=$rep" >> $out
    DIFFPHORE_LIB=$lib python tools/graph_create_cost.py 1 8 40 128 512 2>/dev/null | grep "eager\|replay" >> $out
  done
done
cat $out
