#!/bin/bash
# round 6: fused conv2a+conv2b - parity (dev build A/B) and timing / joules against the two launches
R=$(pwd); O=$R/gpurun_out/r06_i; mkdir -p $O
DEV=$R/superslam_amd/lib/variants/dev.so
timeout 900 python -m pytest tests/test_gpu_alt_paths.py -m gpu -x -q -k "fused_conv2a" -s > $O/pytest_fuse2.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_fuse2.log
tail -25 $O/pytest_fuse2.log
for i in 1 2; do
for mode in split fused; do
  echo "== $mode"
  SSHIP_DEV_LIBRARY=$DEV SUPERSLAM_HIP_CONV2=$mode timeout 300 python bench.py --library $DEV --headline-only --steps 6 --warmup 2 2>&1 | tail -1 | python -c "
import sys, json
b = json.loads(sys.stdin.read()); print(json.dumps({k: b.get(k) for k in ('value', 'ms_per_step', 'power')}))"
done
done | tee $O/ab.txt
