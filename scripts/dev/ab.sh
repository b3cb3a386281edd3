#!/bin/bash
# Developer launcher for one gpurun call (rewritten per experiment; results under gpurun_out/ab/).
set -u
R=$(pwd); O=$R/gpurun_out/ab; mkdir -p $O
timeout 600 python -m pytest tests/test_eigenplaces.py tests/test_gpu_alt_paths.py tests/test_reference_binding.py -m gpu -x -q > $O/pytest_ep.log 2>&1; echo "pytest(ep, alt) rc=$?"; tail -6 $O/pytest_ep.log
python scripts/ep_time.py 50 > $O/ep_time.json 2> $O/ep_time.err; cat $O/ep_time.json; tail -3 $O/ep_time.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_ep; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ep -o ep -- python $R/scripts/ep_time.py 50 --loop-only > $O/ep_under_rocprof.json 2> /tmp/prof_ep.err
DB=$(ls /tmp/prof_ep/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/scripts/rocpd_stats.py $DB --by-grid > $O/ep_kernel_stats.txt || tail -5 /tmp/prof_ep.err > $O/ep_kernel_stats.txt
head -40 $O/ep_kernel_stats.txt
cd $R
timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest(all) rc=$?" | tee -a $O/pytest_gpu.log; tail -6 $O/pytest_gpu.log
