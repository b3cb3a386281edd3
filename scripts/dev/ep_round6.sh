#!/bin/bash
R=$(pwd); O=$R/gpurun_out/r06_h; mkdir -p $O
timeout 900 python -m pytest tests/test_eigenplaces.py tests/test_gpu_alt_paths.py -m gpu -x -q -k "eigenplaces or engine_sizes or descriptor or preprocessing" -s > $O/pytest_ep.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_ep.log
tail -30 $O/pytest_ep.log
for i in 1 2 3; do
python scripts/ep_time.py 200 --loop-only
SSHIP_DEV_LIBRARY=$R/superslam_amd/lib/variants/dev.so SUPERSLAM_HIP_EP_STEM=gemm python scripts/ep_time.py 200 --loop-only
done | tee $O/ep_ab.txt
python scripts/ep_time.py 50 > $O/ep_time.json; cat $O/ep_time.json
cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/prof_ep; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_ep -o ep -- python $R/scripts/ep_time.py 50 --loop-only > /dev/null 2> /tmp/prof_ep.err
DB=$(ls /tmp/prof_ep/*results.db 2>/dev/null | head -1)
[ -n "$DB" ] && python $R/scripts/rocpd_stats.py $DB --by-grid > $O/ep_kernel_stats.txt || tail -5 /tmp/prof_ep.err > $O/ep_kernel_stats.txt
head -30 $O/ep_kernel_stats.txt
