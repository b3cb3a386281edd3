#!/bin/bash
# Wave-cycle breakdown (SQ counters) of every kernel in one bench step.  Two --pmc passes (8 SQ slots per pass),
# kernel-trace only.  usage (GPU box, repo root): scripts/pmc_sq.sh <pairs> <out.txt>
# WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~= WAVE_CYCLES (quad-cycles); VALU_MFMA_BUSY_CYCLES is in cycles.
P=${1:-64}
OUT=${2:-gpurun_out/pmc_sq.txt}
R=$(pwd)
mkdir -p /tmp/pmc_sq gpurun_out
cd /tmp && export TMPDIR=/tmp
: > $R/$OUT
pass() {  # name, counters...
  local n=$1; shift
  timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_sq -o sq_$n -- python $R/bench.py --headline-only --steps 2 --warmup 1 --chunks 2 --pairs $P > /tmp/pmc_sq/run_$n.log 2>&1
  if [ -f /tmp/pmc_sq/sq_${n}_results.db ]; then python $R/scripts/rocpd_pmc.py /tmp/pmc_sq/sq_${n}_results.db >> $R/$OUT; else echo "pass $n failed:" >> $R/$OUT; tail -5 /tmp/pmc_sq/run_$n.log >> $R/$OUT; fi
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES
pass b SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS
