#!/bin/bash
# A/B of the shared layer-0 phore-phore conv (DpBatch.ph_shared + DpWork.t_uniform; DIFFPHORE_PH_SHARE=0 = per pose) on one box:
# configs[3] (one 79-node pharmacophore x ligands x 40 samples) and the one-complex 40-sample call, alternating arms.
export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out
out=gpurun_out/phshare_ab.txt
: > $out
for rep in 1 2; do
  for share in 1 0; do
    v=$(DIFFPHORE_PH_SHARE=$share python bench.py --config 4 --no-cpu-baseline --no-alt --no-e2e --no-other 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])")
    echo "configs[3] share=$share rep=$rep: $v" >> $out
  done
done
for rep in 1 2; do
  for share in 1 0; do
    DIFFPHORE_PH_SHARE=$share python tools/batch_sweep.py --batches 8,20,40 --reps 3 --shared-phore --algos node --out gpurun_out/phshare_sweep_${share}_${rep}.csv > /dev/null 2>&1 < /dev/null
    echo "sweep share=$share rep=$rep" >> $out; cat gpurun_out/phshare_sweep_${share}_${rep}.csv >> $out
  done
done
cat $out
