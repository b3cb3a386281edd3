#!/bin/bash
# round 6: k_lg_ffn4 paired (two tiles per 8-wave workgroup in lock-step, SUPERSLAM_HIP_FFN=42) against two independent 4-wave workgroups per CU
R=$(pwd); DEV=$R/superslam_amd/lib/variants/dev.so; O=$R/gpurun_out/r06_m; mkdir -p $O
for i in 1 2; do
python scripts/dev/stage_energy.py --library $DEV --lg 3,4,5 --calls lg,fe --seconds 1.5 --tag ffn4_$i
SUPERSLAM_HIP_FFN=42 python scripts/dev/stage_energy.py --library $DEV --lg 3,4,5 --calls lg,fe --seconds 1.5 --tag pair_$i
done 2>&1 | grep '^{' | tee -a $O/energy.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    for r in j['rows']: print(j['tag'], r['stage'], r['launch_us'], r['avg_W'], r['sclk_MHz'], r['joules_per_launch'])"
