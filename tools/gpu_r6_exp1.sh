#!/bin/bash
# round 6, experiment 1: (a) the product build with the s0-only instantiation compiled in but unused, (b) timing build DP_S0_TIMING (normals convs without the
# pairs whose harmonics are not e0: wrong results, the dense side's gain), (c) conv-level ablation 11 (cost model of the SH factor on the hidden side)
export PYTHONPATH=$PWD
mkdir -p gpurun_out
B="python bench.py --steps 40 --warmup 5 --no-alt --no-e2e --no-other --no-cpu-baseline"
$B > gpurun_out/exp1_product.json 2> gpurun_out/exp1_product.err
DIFFPHORE_LIB=build_abl/libs0t.so DIFFPHORE_ALLOW_ABLATION=1 $B > gpurun_out/exp1_s0t.json 2> gpurun_out/exp1_s0t.err
run() { python tools/bench_conv.py --algo node --prec f16x3 --mode p2l --iters 50 --no-tail 1 "$@" 2>&1 | tail -1 | python -c "import sys, json; print(round(json.loads(sys.stdin.read())['ms'], 4))"; }
for k in 1 2 3; do
  echo "kind $k base $(run --kind $k) abl11 $(DIFFPHORE_ALLOW_ABLATION=1 run --kind $k --lib build_abl/libabl11.so)"
done > gpurun_out/exp1_abl11.txt 2>&1
tail -c 600 gpurun_out/exp1_product.json; echo; tail -c 600 gpurun_out/exp1_s0t.json; echo; cat gpurun_out/exp1_abl11.txt
