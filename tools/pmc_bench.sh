#!/bin/bash
# rocprofv3 SQ counter passes on bench.py ITSELF (the shipped kernels in their real launches: every tp_conv* kernel of the default
# workload), counters in separate runs with --kernel-trace only.  usage: tools/pmc_bench.sh <tag> [bench args...]
#   -> gpurun_out/<tag>_conv_pmc.csv  (per-kernel averages, first line = kernel source hash the counters belong to)
set -u
tag=${1:-r3}; shift || true
cd "$(dirname "$0")/.." || exit 1
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out
ROOT=$PWD
BARGS="--steps 8 --warmup 2 --no-alt --no-e2e --no-cpu-baseline --no-other $*"
out=gpurun_out/${tag}_conv_pmc.csv
echo "# rocprofv3 --kernel-trace --pmc <set> -- python bench.py $BARGS ; per-kernel averages over all launches; kernel sources $(python -c 'import bench; print(bench.kernel_source_hash())')" > $out
echo "# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES); shader clock = SQ_BUSY_CU_CYCLES / 256 CUs / duration; SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* count quad-cycles" >> $out
echo "kernel,counter,launches,avg_value,avg_duration" >> $out
run() {
  local name=$1; shift
  local d=/tmp/pmcb_${tag}_${name}
  rm -rf "$d"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d "$d" -o r -- python $ROOT/bench.py $BARGS > /tmp/pmcb_${tag}_${name}.out 2>&1)
  local db
  db=$(find "$d" -name '*.db' | head -1)
  if [ -n "$db" ]; then python tools/pmc_dump.py "$db" tp_conv >> $out; else echo "# no db for pass $name" >> $out; tail -3 /tmp/pmcb_${tag}_${name}.out >> $out; fi
}
run a GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU
run c SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
wc -l $out
