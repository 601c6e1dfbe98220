#!/bin/bash
# cycle-stamp breakdown of the edge-side kernel (build_abl/libprofe.so = tools/build_variant.sh profe "-DDP_PROF -DDP_CONV_ONE_TU" dp_conv.hip dp_conv_k0.hip ... dp_conv_k5.hip: the cycle stamps share one device-side counter array, so every kind is instantiated in dp_conv.hip and the per-kind units are empty) in its 16-bit forms
export PYTHONPATH=$PWD DIFFPHORE_ALLOW_ABLATION=1
for prec in f16x3 bf16 bf16x6; do for kind in 2 0; do
  echo "== $prec kind $kind"
  python tools/bench_conv.py --algo edge --prec $prec --kind $kind --lib build_abl/libprofe.so --prof 2>&1 | tail -9
  python tools/bench_conv.py --algo edge --prec $prec --kind $kind 2>&1 | tail -1
done; done
