#!/bin/bash
# one-pair (and small-batch) call: two conv2 launches against the fused kernel with adaptive row segments
R=$(pwd); DEV=$R/superslam_amd/lib/variants/dev.so
for i in 1 2; do
for mode in split fused; do
  echo "== one pair, $mode"; SSHIP_DEV_LIBRARY=$DEV SUPERSLAM_HIP_CONV2=$mode python scripts/dev/latency_loop_dev.py 300 2>&1 | tail -2
done; done
for P in 1 16 32; do for mode in split fused; do
  echo "== $P pairs per call, $mode"; SUPERSLAM_HIP_CONV2=$mode python bench.py --library $DEV --headline-only --no-power --pairs $P --chunks 4 --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print(b['value'], b['ms_per_step'])"
done; done
