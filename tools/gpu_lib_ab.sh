#!/bin/bash
# same-box A/B of the product library against variant builds on the headline step: tools/gpu_lib_ab.sh "<bench args>" lib1.so [lib2.so ...]  ("" = product)
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
args=$1; shift
out=gpurun_out/lib_ab.txt; : > $out
for rep in 1 2 3; do
  for lib in "" "$@"; do
    v=$(DIFFPHORE_LIB=$lib python bench.py $args --no-cpu-baseline --no-alt --no-e2e --no-other 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['value']), d['ms_per_step'])")
    echo "lib=${lib:-product} rep=$rep: $v" >> $out
  done
done
cat $out
