#!/bin/bash
# One --pmc pass (kernel-trace only) over the headline steps; per-kernel counter means.  usage: scripts/pmc_pass.sh <tag> <name-substring> COUNTER...
TAG=$1; SUB=$2; shift 2
R=$(pwd); mkdir -p /tmp/pmc_p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d /tmp/pmc_p -o p_$TAG -- python $R/bench.py --headline-only --steps 2 --warmup 1 --chunks 1 > /tmp/pmc_p/run_$TAG.log 2>&1
if [ -f /tmp/pmc_p/p_${TAG}_results.db ]; then python $R/scripts/rocpd_pmc.py /tmp/pmc_p/p_${TAG}_results.db "$SUB"; else echo "pass $TAG failed"; tail -5 /tmp/pmc_p/run_$TAG.log; fi
