#!/bin/bash
# FETCH_SIZE calibration on k_nms_tile's access pattern (scripts/ubench/fetch_calib.hip): one rocprofv3 --pmc FETCH_SIZE pass, kernel-trace only.
# usage (GPU box, repo root): scripts/fetch_calib.sh [out.json]
set -e
R=$(pwd); OUT=${1:-$R/gpurun_out/fetch_calib.json}; case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
[ -x scripts/ubench/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/fetch_calib scripts/ubench/fetch_calib.hip
mkdir -p /tmp/fcal gpurun_out; cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/fcal -o fc -- $R/scripts/ubench/fetch_calib > /tmp/fcal/run.log 2>&1
python - "$OUT" <<'PY'
import json, sqlite3, sys
exp = json.loads([l for l in open("/tmp/fcal/run.log") if l.startswith("{")][-1])
db = sqlite3.connect("/tmp/fcal/fc_results.db")
cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
ci = {k: i for i, k in enumerate(cols)}
nk = ci.get("kernel_name", ci.get("name", 0))
per = {}
for r in db.execute("select * from counters_collection"):
    if r[ci["counter_name"]] == "FETCH_SIZE":
        per.setdefault(str(r[nk]).split("(")[0], []).append(float(r[ci["value"]]) * 1024.0)
out = {"expected": exp, "kernels": {}}
for k, v in per.items():
    want = exp["k_cells_halo_bytes_requested"] if "halo" in k else exp["buffer_bytes"]
    m = sum(v) / len(v)
    out["kernels"][k] = {"launches": len(v), "FETCH_SIZE_bytes_mean": m, "bytes_requested": want, "FETCH_SIZE_over_requested": round(m / want, 4),
                         "FETCH_SIZE_over_buffer": round(m / exp["buffer_bytes"], 4)}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
