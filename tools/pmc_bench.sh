#!/bin/bash
# rocprofv3 PMC passes on bench.py for one kernel-name filter: tools/pmc_bench.sh <tag> <kernel substring> [bench args]  -> gpurun_out/pmcb_<tag>.txt
set -u
tag=$1; filt=$2; shift 2
cd "$(dirname "$0")/.." || exit 1
export PYTHONPATH=. TMPDIR=/tmp
mkdir -p gpurun_out
: > gpurun_out/pmcb_${tag}.txt
run() {
  local out=/tmp/pmcb_${tag}_$1; shift
  rm -rf "$out"
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d "$out" -o r -- python $OLDPWD/bench.py --steps 4 --warmup 1 --no-alt --no-e2e --no-cpu-baseline --no-other ${ARGS[@]+"${ARGS[@]}"} > /dev/null 2>&1)
  local db; db=$(find "$out" -name '*.db' | head -1)
  [ -n "$db" ] && python tools/pmc_dump.py "$db" "$filt" >> gpurun_out/pmcb_${tag}.txt
}
ARGS=("$@")
run a GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INST_CYCLES_SALU
run c SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_IFETCH SQ_WAIT_IFETCH SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH
cat gpurun_out/pmcb_${tag}.txt
