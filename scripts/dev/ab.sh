#!/bin/bash
# Developer A/B of the LightGlue call on the GPU box (one parametrised launcher instead of round 4's run_[b-z].sh):
#   scripts/dev/ab.sh [-p pairs] [-k keypoints] tag=lib[,ENV=value...] ...
# lib = "default" (the shipped library), "dev" (lib/variants/dev.so) or any lib/variants/<name>.so built with
#   python -m superslam_amd.build --variant <name> -D...      (switches only act in builds with -DSSHIP_DEV_SWITCHES=1)
# The first entry is the reference the others are compared with (matches0 / mscores0 agreement).  Example:
#   gpurun -- 'scripts/dev/ab.sh stream=default res=dev,SUPERSLAM_HIP_ATTN=res nosplit=dev,SUPERSLAM_HIP_LG_SPLIT=1'
# Output: one line per entry (call time, isolated stage times, agreement) + gpurun_out/ab/lg_ab.jsonl.
set -u
P=64; K=600
while getopts "p:k:" o; do case $o in p) P=$OPTARG;; k) K=$OPTARG;; esac; done; shift $((OPTIND - 1))
R=$(pwd); O=$R/gpurun_out/ab; mkdir -p $O; : > $O/lg_ab.jsonl
first=1
for spec in "$@"; do
  tag=${spec%%=*}; rest=${spec#*=}; lib=${rest%%,*}; envs=""; [ "$rest" != "$lib" ] && envs=$(echo "${rest#*,}" | tr ',' ' ')
  case $lib in default) so=$R/superslam_amd/lib/libsuperslam_hip.so;; *) so=$R/superslam_amd/lib/variants/$lib.so;; esac
  if [ $first = 1 ]; then extra="--save /tmp/ab_ref.npz"; first=0; else extra="--ref /tmp/ab_ref.npz"; fi
  env SSHIP_DEV_LIBRARY=$so $envs python scripts/dev/lg_ab.py --pairs $P --kp $K --tag $tag $extra 2>/dev/null | tail -1 >> $O/lg_ab.jsonl
done
python - <<'PY'
import json
for l in open("gpurun_out/ab/lg_ab.jsonl"):
    try: b = json.loads(l)
    except Exception: print("BAD", l[:300]); continue
    print(b["tag"], "call", b["call_ms_median"], "min", b["call_ms_min"], {k: round(v * 1e3, 1) for k, v in b["stage_ms"].items()}, b.get("vs_ref"))
PY
