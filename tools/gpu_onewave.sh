#!/bin/bash
# one wave per SIMD with the whole register file (build_abl/libonewave.so: -DDP_WG_PER_CU=1 on the layer-3 node-side kernels) vs the product
export PYTHONPATH=$PWD
for lib in "" build_abl/libonewave.so; do
  echo "== lib: ${lib:-product}"
  for prec in f16x3 bf16; do
    DIFFPHORE_LIB=$lib python tools/bench_conv.py --algo node --prec $prec --kind 3 --no-tail 2>&1 | tail -1
    DIFFPHORE_LIB=$lib python tools/bench_conv.py --algo node --prec $prec --kind 3 2>&1 | tail -1
  done
done
