#!/bin/bash
# rocprofv3 PMC passes on the fused conv micro-benchmark (tools/bench_conv.py); counters in separate runs, no tracing besides kernel-trace.
# usage: tools/pmc_conv.sh <tag> <bench_conv args...>      -> gpurun_out/pmc_<tag>_{a,b,c}.txt (per-kernel averages)
set -u
tag=$1; shift
export PYTHONPATH=.
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
run() {
  local name=$1; shift
  local out=/tmp/pmc_${tag}_${name}
  rm -rf "$out"
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d "$out" -o r -- python tools/bench_conv.py --iters 5 "${ARGS[@]}" > /dev/null 2>&1
  local db
  db=$(find "$out" -name '*.db' | head -1)
  if [ -n "$db" ]; then python tools/pmc_dump.py "$db" tp_conv > gpurun_out/pmc_${tag}_${name}.txt; else echo "no db for $name" > gpurun_out/pmc_${tag}_${name}.txt; fi
}
ARGS=("$@")
run a GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run b SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU
run c SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT
cat gpurun_out/pmc_${tag}_*.txt
