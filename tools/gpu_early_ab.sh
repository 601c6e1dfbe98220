#!/bin/bash
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/early_ab.txt; : > $out
for rep in 1 2; do
  for lib in build_abl/libearly0.so build_abl/libearly4k.so; do
    for P in 64 96 128 192 256; do
      v=$(DIFFPHORE_LIB=$lib python tools/graph_timeline_run.py $P 2 2>/dev/null | tail -1)
      echo "lib=$lib $v" >> $out
    done
  done
done
cat $out
