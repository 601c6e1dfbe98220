export PYTHONPATH=.
timeout 600 python -m pytest tests/test_conv_gpu.py -x -q -m gpu -k node 2>&1 | tail -3
for v in nodeBase node0; do for k in 3 2 1 0; do timeout 120 python tools/bench_conv.py --kind $k --algo node --lib build_abl/libabl_$v.so 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], d['kind'], round(d['ms'],4))" $v; done; done
