#!/bin/bash
# HBM traffic of every kernel of the headline step from PMC counters, per MI355X_MICROARCH.md "HBM": FETCH_SIZE and
# WRITE_SIZE in SEPARATE --pmc passes (TCC slots), kernel-trace only; FETCH_SIZE is doubled on gfx950 for wide coalesced
# reads.  The profiled command is `bench.py --headline-only`: EVERY launch in the trace is a headline launch of <pairs>
# pairs (round 1 averaged the 2-image latency-loop launches into the mean - VERDICT r01 "What's weak" 6).
# usage (on the GPU box, from the repo root): [PMC_EXTRA="--library lib/variants/x.so"] scripts/pmc_traffic.sh <pairs> [out.json]
set -e
P=${1:-64}
R=$(pwd)
OUT=${2:-$R/gpurun_out/pmc_traffic.json}; case $OUT in /*) ;; *) OUT=$R/$OUT;; esac
mkdir -p /tmp/pmc_t gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_t -o t_$c -- python $R/bench.py --headline-only --no-power --steps 2 --warmup 1 --chunks 2 --pairs $P $PMC_EXTRA > /tmp/pmc_t/run_$c.log 2>&1
done
python - "$R" "$P" "$OUT" <<'PY'
import hashlib, json, sqlite3, sys, time
root, pairs, outp = sys.argv[1], int(sys.argv[2]), sys.argv[3]
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    db = sqlite3.connect(f"/tmp/pmc_t/t_{c}_results.db")
    cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
    ci = {k: i for i, k in enumerate(cols)}
    nk = ci.get("kernel_name", ci.get("name", 0))
    for r in db.execute("select * from counters_collection"):
        if r[ci["counter_name"]] != c:
            continue
        name = str(r[nk])
        d = per.setdefault(name, {})
        d.setdefault(c, []).append(float(r[ci["value"]]))
kernels = {}
for name, d in per.items():
    f, w = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
    fm, wm = (sum(f) / len(f) if f else 0.0), (sum(w) / len(w) if w else 0.0)
    short = name.split("(")[0][:110]
    # gfx950: FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md) -> x2; WRITE_SIZE as is
    kernels[short] = {"launches": max(len(f), len(w)), "FETCH_SIZE_KB_mean": fm, "WRITE_SIZE_KB_mean": wm,
                      "hbm_bytes_per_launch": (2.0 * fm + wm) * 1024.0}
out = {"pairs_per_call": pairs, "images_per_launch": 2 * pairs, "headline_launches_only": True,
       # provenance (bench.py prints it next to roofline.traffic: the value is read from this file, not measured in the bench run)
       "collected_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
       "conv_pp_sha16": hashlib.sha256(open(f"{root}/superslam_amd/csrc/conv_pp.hip", "rb").read()).hexdigest()[:16],
       "command": f"bench.py --headline-only --steps 2 --warmup 1 --chunks 2 --pairs {pairs}",
       "note": "bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch (mean over the launches of the trace, all of them headline "
               "launches); separate --pmc passes", "kernels": kernels}
dom = [k for k in kernels if "conv3x3_pp<64, 64, true, true>" in k]
if dom:
    out["kernel"] = dom[0]
    out.update({k: kernels[dom[0]][k] for k in ("FETCH_SIZE_KB_mean", "WRITE_SIZE_KB_mean", "hbm_bytes_per_launch", "launches")})
json.dump(out, open(outp, "w"), indent=1)
json.dump({k: v for k, v in out.items() if k != "kernels"}, open(f"{root}/gpurun_out/pmc_conv1ab.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))
for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches"])[:14]:
    print(f"{v['launches']:4d} x {v['hbm_bytes_per_launch'] / 1e6:10.2f} MB  (fetch x2 {2 * v['FETCH_SIZE_KB_mean'] / 1e3:9.2f} MB, write {v['WRITE_SIZE_KB_mean'] / 1e3:9.2f} MB)  {k[:90]}")
PY
