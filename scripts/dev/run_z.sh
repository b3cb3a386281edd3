#!/bin/bash
R=$(pwd); O=$R/gpurun_out/z; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_alt_paths.py tests/test_gpu_bench_batch_parity.py tests/test_gpu_parity.py tests/test_gpu_stress_shapes.py tests/test_gpu_lifetime_streams.py -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_p1; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_p1 -o ks -- python $R/bench.py --headline-only --steps 3 --warmup 1 --pairs 1 --chunks 50 > $O/bench_p1.json 2> /tmp/prof_p1.err
DB=$(ls /tmp/prof_p1/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/scripts/rocpd_stats.py $DB --by-grid | cut -c1-200 > $O/kernel_stats_P1.txt
grep -i "desc_head\|total kernel\|topk" $O/kernel_stats_P1.txt; cut -c1-120 $O/bench_p1.json
