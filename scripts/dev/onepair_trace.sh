#!/bin/bash
# per-kernel table of the one-pair call (rocprofv3 --kernel-trace of scripts/latency_loop.py)
R=$(pwd); O=$R/gpurun_out/r06_k; mkdir -p $O
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_1p
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_1p -o p1 -- python $R/scripts/latency_loop.py 100 > $O/latency_loop.txt 2> /tmp/prof_1p.err
DB=$(ls /tmp/prof_1p/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/scripts/rocpd_stats.py $DB --by-grid > $O/onepair_kernel_stats.txt || tail -5 /tmp/prof_1p.err
cd $R; cat $O/latency_loop.txt; head -50 $O/onepair_kernel_stats.txt
