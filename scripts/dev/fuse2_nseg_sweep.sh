#!/bin/bash
R=$(pwd); DEV=$R/superslam_amd/lib/variants/dev.so
for rep in 1 2; do for n in 1 2 3 4 6; do
SUPERSLAM_HIP_CONV2_NSEG=$n python scripts/dev/stage_energy.py --library $DEV --sp 15 --seconds 1.0 --tag nseg$n 2>&1 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    for r in j['rows']: print(j['tag'], r['launch_us'], r['avg_W'], r['sclk_MHz'], r['joules_per_launch'])"
done; done
