#!/bin/bash
# one-pair call: per-kernel durations with the alternative 128-input-channel conv kernels
R=$(pwd); O=$R/gpurun_out/p1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in th8 ct32; do
  rm -rf /tmp/prof_p1; SUPERSLAM_HIP_CONV128=$v timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p1 -o ks -- python $R/bench.py --headline-only --steps 3 --warmup 1 --pairs 1 --chunks 50 > $O/bench_p1_$v.json 2> /tmp/prof_p1.err
  DB=$(ls /tmp/prof_p1/*results.db 2>/dev/null | head -1)
  [ -n "$DB" ] && python $R/scripts/rocpd_stats.py $DB --by-grid | grep -i "conv\|total" | cut -c1-200 > $O/kernel_stats_P1_$v.txt
  echo "== $v"; cat $O/kernel_stats_P1_$v.txt; cut -c1-120 $O/bench_p1_$v.json
done
