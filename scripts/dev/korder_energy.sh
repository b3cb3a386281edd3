#!/bin/bash
# round 6: (kx, k-step, ky) order with shared row fragments against the tap-major order (variant korder0), joules per launch
R=$(pwd); O=$R/gpurun_out/r06_j; mkdir -p $O
[ -f $R/superslam_amd/lib/variants/korder0.so ] || python -m superslam_amd.build --variant korder0 -DSSHIP_K_ROWSHARE=0   # the tap-major build (hipcc on the box: ~2 min)
for i in 1 2; do
python scripts/dev/stage_energy.py --library $R/superslam_amd/lib/variants/korder0.so --sp 1,15,4 --calls fe --seconds 1.5 --tag tapmajor_$i
python scripts/dev/stage_energy.py --library $R/superslam_amd/lib/libsuperslam_hip.so --sp 1,15,4 --calls fe --seconds 1.5 --tag rowshare_$i
done 2>&1 | grep '^{' | tee -a $O/energy.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    for r in j['rows']: print(j['tag'], r['stage'], r['launch_us'], r['avg_W'], r['sclk_MHz'], r['joules_per_launch'])"
