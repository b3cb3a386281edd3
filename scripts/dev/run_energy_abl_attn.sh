#!/bin/bash
# Energy ablation of the attention kernel (self-attention launch of a 64-pair call): variant builds -DSSHIP_ATTN_ABL=<n>, rocm-smi power x launch time
mkdir -p gpurun_out
O=gpurun_out/energy_abl_attn.txt
: > $O
run() { tag=$1; shift; bash scripts/dev/power_poll.sh $tag "$@" >> $O 2>&1; tail -1 /tmp/pp_$tag.log | sed "s/^/$tag /" >> $O; }
run base python scripts/dev/loop_kernel.py attn
for v in 1 2 3; do
  SUPERSLAM_HIP_LIBRARY=$(pwd)/superslam_amd/lib/variants/attnabl$v.so run abl$v python scripts/dev/loop_kernel.py attn
done
python - <<'PY'
import re
txt = open("gpurun_out/energy_abl_attn.txt").read()
rows, ms = {}, {}
for line in txt.splitlines():
    m = re.match(r"^([a-zA-Z0-9_]+): .*sclk clock level: 1: \((\d+)Mhz\).*Power \(W\): ([0-9.]+)", line)
    if m: rows.setdefault(m.group(1), []).append((int(m.group(2)), float(m.group(3))))
    m = re.match(r"^([a-zA-Z0-9_]+) \w+ avg_ms ([0-9.]+)", line)
    if m: ms[m.group(1)] = float(m.group(2))
names = {"base": "baseline", "abl1": "no MFMAs", "abl2": "no exp2 in the key loop", "abl3": "neither"}
for k, v in rows.items():
    p = sorted(x[1] for x in v)[len(v) // 2]; c = sorted(x[0] for x in v)[len(v) // 2]
    t = ms.get(k, float("nan"))
    print(f"{k:6s} {names.get(k, ''):26s} sclk {c:5d} MHz  power {p:6.0f} W  launch {t * 1e3:7.1f} us  energy {p * t:7.2f} mJ  cycles {c * t:7.0f} k")
PY
