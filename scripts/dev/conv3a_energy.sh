#!/bin/bash
# conv3a: conv3x3_pp<64,64> with two cout tiles (shipped) against the rolling-window one-layer kernel (developer build, SUPERSLAM_HIP_CONV3A=roll)
R=$(pwd); DEV=$R/superslam_amd/lib/variants/dev.so; O=$R/gpurun_out/r06_l; mkdir -p $O
for i in 1 2; do
python scripts/dev/stage_energy.py --library $DEV --sp 4 --calls fe --seconds 1.5 --tag pp_$i
SUPERSLAM_HIP_CONV3A=roll python scripts/dev/stage_energy.py --library $DEV --sp 4 --calls fe --seconds 1.5 --tag roll_$i
done 2>&1 | grep '^{' | tee -a $O/energy.jsonl | python -c "
import sys, json
for l in sys.stdin:
    j = json.loads(l)
    for r in j['rows']: print(j['tag'], r['stage'], r['launch_us'], r['avg_W'], r['sclk_MHz'], r['joules_per_launch'])"
for mode in pp roll; do
  echo "== one pair, conv3a $mode"; SSHIP_DEV_LIBRARY=$DEV SUPERSLAM_HIP_CONV3A=$mode python scripts/dev/latency_loop_dev.py 300 2>&1 | tail -1
done
