#!/bin/bash
R=$(pwd); DEV=$R/superslam_amd/lib/variants/dev.so
for i in 1 2; do
python scripts/dev/stage_energy.py --library $DEV --lg 1,2 --calls lg --seconds 1.5 --tag stream_$i
SUPERSLAM_HIP_ATTN=res python scripts/dev/stage_energy.py --library $DEV --lg 1,2 --calls lg --seconds 1.5 --tag res_$i
done 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    for r in j['rows']: print(j['tag'], r['stage'], r['launch_us'], r['avg_W'], r['sclk_MHz'], r['joules_per_launch'])"
