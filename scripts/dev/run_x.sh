#!/bin/bash
# k_lg_ffn4 stagger sweep: LightGlue call (64 pairs, default two-stream split) and isolated FFN stage times
for st in 0 6000 10000 14000 20000 28000; do
  SSHIP_FFN4_STAGGER=$st python scripts/dev/lg_ab.py --pairs 64 --tag stagger_$st 2>&1 | tail -1 | python -c "
import sys, json
b = json.loads(sys.stdin.read())
print(b['tag'], 'call', b['call_ms_median'], 'min', b['call_ms_min'], 'ffn', b['stage_ms']['ffn_self+proj'], b['stage_ms']['ffn_cross+proj'], b['stage_ms']['ffn_last+final'], 'checksum', b['checksum'])"
done
