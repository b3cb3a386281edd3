#!/bin/bash
# Developer A/B launcher for one gpurun call (rewritten per experiment; results under gpurun_out/ab/).
set -u
O=gpurun_out/ab; mkdir -p $O
run() { tag=$1; shift; env "$@" python scripts/dev/lg_ab.py --pairs ${P:-64} --kp ${KP:-600} --tag $tag ${EXTRA:-} 2>&1 | tail -1 >> $O/lg_ab.jsonl; }
: > $O/lg_ab.jsonl
EXTRA="--save /tmp/res.npz" run res A=1
EXTRA="--ref /tmp/res.npz" run stream SUPERSLAM_HIP_ATTN=stream
EXTRA="--ref /tmp/res.npz" run res_nosplit SUPERSLAM_HIP_LG_SPLIT=1
EXTRA="--ref /tmp/res.npz" run stream_nosplit SUPERSLAM_HIP_ATTN=stream SUPERSLAM_HIP_LG_SPLIT=1
KP=1024 EXTRA="--save /tmp/res1024.npz" run res_1024 A=1
KP=1024 EXTRA="--ref /tmp/res1024.npz" run stream_1024 SUPERSLAM_HIP_ATTN=stream
P=16 EXTRA="--save /tmp/res16.npz" run res_p16 A=1
P=16 EXTRA="--ref /tmp/res16.npz" run stream_p16 SUPERSLAM_HIP_ATTN=stream
python - <<'PY'
import json
for l in open("gpurun_out/ab/lg_ab.jsonl"):
    try: b = json.loads(l)
    except Exception: print("BAD", l[:300]); continue
    print(b["tag"], "call", b["call_ms_median"], "min", b["call_ms_min"], {k: round(v*1e3,1) for k, v in b["stage_ms"].items()}, b["checksum"], b.get("vs_ref"))
PY
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log; tail -5 $O/pytest_gpu.log
timeout 300 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json | head -c 3000
