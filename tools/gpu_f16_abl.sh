#!/bin/bash
# timing ablations of the f16x3 node-side kernel (layer-3 phore->ligand conv, 512 poses): build_abl/libfablN.so = -DDPF_ABL=N builds of dp_conv_node_f16_k3.hip
export PYTHONPATH=$PWD
run() { python tools/bench_conv.py --algo node --prec f16x3 --kind 3 --mode p2l --iters 50 "$@" 2>&1 | tail -1 | python -c "import sys, json; print(round(json.loads(sys.stdin.read())['ms'], 4))"; }
echo "base $(run)"
for n in 1 2 3 4 5 6 7 8 9 10; do
  [ -f build_abl/libfabl$n.so ] && echo "abl$n $(DIFFPHORE_LIB=build_abl/libfabl$n.so run)"
done
