#!/bin/bash
# Energy ablation of the throughput FFN kernel (k_lg_ffn4): rocm-smi power x launch time of the SelfBlock FFN stage for variant builds
# (python superslam_amd/build.py --variant ffn4abl<n> -DSSHIP_FFN4_ABL=<n>).  Results of the variants are wrong by design.
mkdir -p gpurun_out
O=gpurun_out/energy_abl.txt
: > $O
run() { tag=$1; shift; bash scripts/dev/power_poll.sh $tag "$@" >> $O 2>&1; tail -1 /tmp/pp_$tag.log | sed "s/^/$tag /" >> $O; }
run base python scripts/dev/loop_kernel.py ffn
for v in 1 2 4 8 5 3; do
  SUPERSLAM_HIP_LIBRARY=$(pwd)/superslam_amd/lib/variants/ffn4abl$v.so run abl$v python scripts/dev/loop_kernel.py ffn
done
python - <<'PY'
import re
txt = open("gpurun_out/energy_abl.txt").read()
rows, ms = {}, {}
for line in txt.splitlines():
    m = re.match(r"^([a-zA-Z0-9]+): .*sclk clock level: 1: \((\d+)Mhz\).*Power \(W\): ([0-9.]+)", line)
    if m: rows.setdefault(m.group(1), []).append((int(m.group(2)), float(m.group(3))))
    m = re.match(r"^([a-zA-Z0-9]+) ffn avg_ms ([0-9.]+)", line)
    if m: ms[m.group(1)] = float(m.group(2))
names = {"base": "baseline", "abl1": "no MFMAs", "abl2": "no LayerNorm / GELU math", "abl4": "no weight stream (fragments of offset 0)", "abl8": "no projection epilogue",
         "abl5": "no MFMAs, no weight stream", "abl3": "no MFMAs, no LayerNorm / GELU math"}
for k, v in rows.items():
    p = sorted(x[1] for x in v)[len(v) // 2]; c = sorted(x[0] for x in v)[len(v) // 2]
    t = ms.get(k, float("nan"))
    print(f"{k:6s} {names.get(k, ''):42s} sclk {c:5d} MHz  power {p:6.0f} W  launch {t * 1e3:7.1f} us  energy {p * t:7.2f} mJ")
PY
