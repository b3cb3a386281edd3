#!/bin/bash
# joules per launch of conv2a, conv2b+pool (two launches) and the fused kernel, plus the whole front-end call in both modes
R=$(pwd); O=$R/gpurun_out/r06_i; mkdir -p $O
DEV=$R/superslam_amd/lib/variants/dev.so
LIB=${1:-$R/superslam_amd/lib/libsuperslam_hip.so}
for i in 1 2; do
python scripts/dev/stage_energy.py --library $LIB --sp 2,3,15 --seconds 1.5 --tag layers_$i
SUPERSLAM_HIP_CONV2=split python scripts/dev/stage_energy.py --library $DEV --calls fe --seconds 2 --tag fe_split_$i
SUPERSLAM_HIP_CONV2=fused python scripts/dev/stage_energy.py --library $DEV --calls fe --seconds 2 --tag fe_fused_$i
done 2>&1 | grep '^{' | tee -a $O/energy.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    for r in j['rows']: print(j['tag'], r['stage'], r['launch_us'], r['avg_W'], r['sclk_MHz'], r['joules_per_launch'])"
