#!/bin/bash
# headline vs pairs per library call (same box, same run)
for p in 64 96 128; do
  c=$((512 / p)); [ $p = 96 ] && c=5
  python bench.py --headline-only --pairs $p --chunks $c --steps 10 --warmup 2 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('pairs/call', b['config']['pairs_per_call'], 'calls/step', b['config']['calls_per_step'], 'value', b['value'], 'ms/step', b['ms_per_step'])"
done
