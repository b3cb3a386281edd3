#!/bin/bash
# k_lg_ffn4 phase trace of tile iteration 0 / 1 / 2 in a ONE-stream 64-pair call (1 216 tiles on 512 workgroup slots)
for it in 0 1 2; do
  echo "== tile iteration $it"
  SUPERSLAM_HIP_LG_SPLIT=1 SSHIP_FFN_TRACE=1 SSHIP_FFN_TRACE_IT=$it python scripts/dev/lg_ab.py --pairs 64 --reps 2 2>&1 | grep "ffn4 trace" | sed -n "9,10p"
done
SUPERSLAM_HIP_LG_SPLIT=1 python scripts/dev/lg_ab.py --pairs 64 --tag one_stream 2>&1 | tail -1 | cut -c1-330
python scripts/dev/lg_ab.py --pairs 64 --tag two_streams 2>&1 | tail -1 | cut -c1-330
