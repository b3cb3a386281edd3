#!/bin/bash
for st in 0 14000; do
  echo "== stagger $st, tile iteration 0 / 1"
  for it in 0 1; do
    SSHIP_FFN4_STAGGER=$st SUPERSLAM_HIP_LG_SPLIT=1 SSHIP_FFN_TRACE=1 SSHIP_FFN_TRACE_IT=$it python scripts/dev/lg_ab.py --pairs 64 --reps 2 2>&1 | grep "ffn4 trace" | sed -n "9,10p"
  done
done
