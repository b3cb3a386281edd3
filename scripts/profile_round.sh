#!/bin/bash
# One-shot evidence collection on the GPU box (repo root): tests, bench, rocprofv3 kernel stats of the headline steps,
# HBM traffic + SQ counter passes, phase traces.  Everything lands under gpurun_out/<tag>/ as small text files.
# usage: scripts/profile_round.sh <tag> [pairs] [skip-tests]
TAG=${1:-round}; P=${2:-64}; SKIPT=${3:-}
R=$(pwd); O=$R/gpurun_out/$TAG; mkdir -p $O
if [ -z "$SKIPT" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
  cp gpurun_out/parity_report.json $O/ 2>/dev/null
fi
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"   # roofline.traffic measured in this very run (child rocprofv3 passes)
cd /tmp && export TMPDIR=/tmp
# kernel stats of the SAME command's headline part only: every launch in this trace belongs to a timed-region-shaped step
rm -rf /tmp/prof_ks; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o ks -- python $R/bench.py --headline-only --steps 3 --warmup 1 --pairs $P > $O/bench_under_rocprof.json 2> /tmp/prof_ks.err
DB=$(ls /tmp/prof_ks/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/scripts/rocpd_stats.py $DB --by-grid > $O/kernel_stats_P$P.txt || tail -5 /tmp/prof_ks.err > $O/kernel_stats_P$P.txt
cd $R
timeout 400 scripts/pmc_traffic.sh $P $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1; cp gpurun_out/pmc_conv1ab.json $O/ 2>/dev/null
timeout 400 scripts/pmc_sq.sh $P gpurun_out/$TAG/pmc_sq_raw.txt; python scripts/pmc_sq_table.py $O/pmc_sq_raw.txt > $O/pmc_sq_P$P.txt; rm -f $O/pmc_sq_raw.txt
SSHIP_FFN_TRACE=1 timeout 120 python bench.py --library $R/superslam_amd/lib/variants/dev.so --headline-only --steps 1 --warmup 1 --chunks 1 --pairs $P 2>&1 | grep -E "ffn4? trace" | sed -n "19,22p" > $O/ffn_phase_trace.txt
# EigenPlaces: per-descriptor latency + the rocprofv3 kernel table of the device-resident loop
python scripts/ep_time.py 50 > $O/ep_time.json 2> /dev/null
cd /tmp; rm -rf /tmp/prof_ep; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ep -o ep -- python $R/scripts/ep_time.py 50 --loop-only > /dev/null 2> /tmp/prof_ep.err
DB=$(ls /tmp/prof_ep/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/scripts/rocpd_stats.py $DB --by-grid > $O/ep_kernel_stats.txt || tail -5 /tmp/prof_ep.err > $O/ep_kernel_stats.txt
cd $R
ls -la $O
