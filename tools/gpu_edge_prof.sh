#!/bin/bash
# cycle-stamp breakdown of the edge-side kernel (build_abl/libprofe.so = -DDP_PROF on dp_conv.hip) in its 16-bit forms
export PYTHONPATH=$PWD DIFFPHORE_ALLOW_ABLATION=1
for prec in f16x3 bf16 bf16x6; do for kind in 2 0; do
  echo "== $prec kind $kind"
  python tools/bench_conv.py --algo edge --prec $prec --kind $kind --lib build_abl/libprofe.so --prof 2>&1 | tail -9
  python tools/bench_conv.py --algo edge --prec $prec --kind $kind 2>&1 | tail -1
done; done
