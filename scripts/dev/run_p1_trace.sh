#!/bin/bash
# per-kernel durations of the ONE-pair library call (single_pair_protocol's unit): rocprofv3 kernel trace of bench.py --headline-only --pairs 1
R=$(pwd); O=$R/gpurun_out/p1; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_p1; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p1 -o ks -- python $R/bench.py --headline-only --steps 3 --warmup 1 --pairs 1 --chunks 50 > $O/bench_p1.json 2> /tmp/prof_p1.err
DB=$(ls /tmp/prof_p1/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/scripts/rocpd_stats.py $DB --by-grid > $O/kernel_stats_P1.txt || tail -5 /tmp/prof_p1.err > $O/kernel_stats_P1.txt
cd $R
python scripts/dev/lg_ab.py --pairs 1 --tag p1 > $O/lg_p1.txt 2>&1
tail -3 $O/bench_p1.json | cut -c1-600
