export PYTHONPATH=$PWD TMPDIR=/tmp
python -m pytest tests/test_conv_bf16_gpu.py tests/test_conv_gpu.py tests/test_conv_f16_gpu.py -q -x 2>&1 | tail -3
for prec in bf16 f16x3 f32; do for kind in 2 3; do
  for lib in base build_abl/libnopipe.so; do
    if [ "$lib" == "base" ]; then L=""; else L="--lib $lib"; fi
    echo "$prec kind $kind $lib: $(python tools/bench_conv.py --algo edge --prec $prec --kind $kind --iters 50 $L 2>&1 | tail -1 | cut -c1-170)"
  done
done; done
for cfg in 5 2; do
for v in base build_abl/libnopipe.so base build_abl/libnopipe.so; do
  if [ "$v" == "base" ]; then unset DIFFPHORE_LIB; else export DIFFPHORE_LIB=$v; fi
  python bench.py --config $cfg --steps 60 --warmup 5 --no-alt --no-e2e --no-cpu-baseline --no-other 2>/dev/null | python -c "import sys, json; d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg$cfg $v', round(d['value']), round(d['ms_per_step'], 4))"
done; done
