#!/bin/bash
# round 3, GPU call 1: validation of the measurement / parity batch + attention softmax variants + single-pair timeline
R=$(pwd); O=$R/gpurun_out/r03_a; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
cp gpurun_out/parity_report.json $O/ 2>/dev/null
for v in 1 2 3; do
  SUPERSLAM_HIP_ATTN_V=$v timeout 300 python scripts/lg_stage_times.py 64 600 2>&1 | tail -1 | sed "s/^/attn_v$v /" >> $O/attn_variants.txt
  SUPERSLAM_HIP_ATTN_V=$v timeout 300 python scripts/lg_call_time.py 64 600 20 2>&1 | tail -1 | sed "s/^/attn_v$v /" >> $O/attn_variants.txt
  SUPERSLAM_HIP_ATTN_V=$v timeout 300 python scripts/lg_stage_times.py 1 600 2>&1 | tail -1 | sed "s/^/attn_v$v P=1 /" >> $O/attn_variants.txt
done
for v in 2 3; do
  SUPERSLAM_HIP_ATTN_V=$v timeout 600 python -m pytest tests/test_gpu_lightglue_layers.py tests/test_gpu_parity.py -m gpu -x -q -k "lightglue or lg or LG or match" > $O/pytest_lg_attn_v$v.log 2>&1; echo "attn v$v pytest rc=$?" | tee -a $O/attn_variants.txt
done
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 120 python scripts/latency_loop.py 60 600 > $O/latency.txt 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ks; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ks -o ks -- python $R/bench.py --headline-only --steps 3 --warmup 1 --pairs 64 > $O/bench_under_rocprof.json 2> /tmp/prof_ks.err
DB=$(ls /tmp/prof_ks/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/scripts/rocpd_stats.py $DB --by-grid > $O/kernel_stats_P64.txt || tail -5 /tmp/prof_ks.err > $O/kernel_stats_P64.txt
rm -rf /tmp/prof_l; timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/prof_l -o l -- python $R/scripts/latency_loop.py 20 600 > $O/latency_under_rocprof.txt 2> /tmp/prof_l.err
DB=$(ls /tmp/prof_l/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/scripts/rocpd_stats.py $DB --by-grid > $O/kernel_stats_P1.txt || tail -5 /tmp/prof_l.err > $O/kernel_stats_P1.txt
cd $R
timeout 400 scripts/pmc_traffic.sh 64 $O/pmc_traffic.json > $O/pmc_traffic.log 2>&1; cp gpurun_out/pmc_conv1ab.json $O/ 2>/dev/null
ls -la $O
