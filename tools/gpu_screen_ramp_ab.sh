export PYTHONPATH=$PWD
for r in 1 0 1 0; do
  DIFFPHORE_SCREEN_RAMP=$r python bench.py --config 4 --steps 5 --warmup 2 --no-alt --no-cpu-baseline --no-other 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); e = d['config']['e2e_screen']
print('ramp $r', round(e['value']), round(e['steady_value']), e['first_batch_delivered_s'], round(e['seconds'],3), e['host_unit_build_s'])"
done
