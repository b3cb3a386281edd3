#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters, per MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE in
# SEPARATE --pmc passes (TCC slots), kernel-trace only; FETCH_SIZE is doubled on gfx950 for wide coalesced reads.
# usage (on the GPU box, from the repo root): scripts/pmc_traffic.sh <pairs>
set -e
P=${1:-64}
R=$(pwd)
mkdir -p /tmp/pmc_t gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_t -o t_$c -- python $R/bench.py --steps 3 --warmup 1 --pairs $P --no-cpu-baseline > /tmp/pmc_t/run_$c.log 2>&1
done
python - "$R" "$P" <<'PY'
import json, sqlite3, sys
root, pairs = sys.argv[1], int(sys.argv[2])
out = {"pairs_per_step": pairs, "images_per_launch": 2 * pairs, "kernel": "conv3x3_pp<64, 64, true, true>"}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = sqlite3.connect(f"/tmp/pmc_t/t_{c}_results.db")
    cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
    ci = {k: i for i, k in enumerate(cols)}
    vals = [float(r[ci["value"]]) for r in db.execute("select * from counters_collection")
            if "conv3x3_pp<64, 64, true, true>" in str(r[ci.get("kernel_name", ci.get("name", 0))]) and r[ci["counter_name"]] == c]
    out[c + "_KB_mean"] = sum(vals) / max(1, len(vals))
    out[c + "_launches"] = len(vals)
# gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md) -> x2; WRITE_SIZE as is
out["hbm_bytes_per_launch"] = (2.0 * out["FETCH_SIZE_KB_mean"] + out["WRITE_SIZE_KB_mean"]) * 1024.0
out["note"] = "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; separate --pmc passes; algorithmic bytes = u8 image read + pooled fp16 output"
json.dump(out, open(f"{root}/gpurun_out/pmc_conv1ab.json", "w"), indent=1)
print(json.dumps(out))
PY
