#!/bin/bash
# A/B of the software-pipelined one-product edge-side stage (StageP, dp_conv_impl.h) against the unpipelined build (tools/build_variant.sh nopipe
# "-DDP_PIPE_NS1=0" dp_conv_k0.hip ... dp_conv_k5.hip): the bf16 conv tests, the conv alone at the configs[4] phore-phore shape, the configs[4] step
export PYTHONPATH=$PWD TMPDIR=/tmp
python -m pytest tests/test_conv_bf16_gpu.py tests/test_conv_gpu.py -q -x 2>&1 | tail -3
for kind in 0 1 2 3; do
  for lib in base build_abl/libnopipe.so; do
    if [ "$lib" == "base" ]; then L=""; else L="--lib $lib"; fi
    echo "kind $kind $lib: $(python tools/bench_conv.py --algo edge --prec bf16 --kind $kind --iters 50 $L 2>&1 | tail -1)"
  done
done
for v in base build_abl/libnopipe.so base build_abl/libnopipe.so; do
  if [ "$v" == "base" ]; then unset DIFFPHORE_LIB; else export DIFFPHORE_LIB=$v; fi
  python bench.py --config 5 --steps 60 --warmup 5 --no-alt --no-e2e --no-cpu-baseline --no-other 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg5 $v', round(d['value']), round(d['ms_per_step'], 4))"
done
