#!/bin/bash
# Energy ablation of the dominant kernel (conv1a + conv1b + pool, conv_pp.hip) and of conv2a: variant builds -DSSHIP_PP_ABL=<n>, rocm-smi power x launch time
mkdir -p gpurun_out
O=gpurun_out/energy_abl_conv.txt
: > $O
run() { tag=$1; shift; bash scripts/dev/power_poll.sh $tag "$@" >> $O 2>&1; tail -1 /tmp/pp_$tag.log | sed "s/^/$tag /" >> $O; }
for st in conv1ab conv2a; do
  run base_$st python scripts/dev/loop_kernel.py $st
  for v in 1 2 3; do
    SUPERSLAM_HIP_LIBRARY=$(pwd)/superslam_amd/lib/variants/ppabl$v.so run abl${v}_$st python scripts/dev/loop_kernel.py $st
  done
done
python - <<'PY'
import re
txt = open("gpurun_out/energy_abl_conv.txt").read()
rows, ms = {}, {}
for line in txt.splitlines():
    m = re.match(r"^([a-zA-Z0-9_]+): .*sclk clock level: 1: \((\d+)Mhz\).*Power \(W\): ([0-9.]+)", line)
    if m: rows.setdefault(m.group(1), []).append((int(m.group(2)), float(m.group(3))))
    m = re.match(r"^([a-zA-Z0-9_]+) \w+ avg_ms ([0-9.]+)", line)
    if m: ms[m.group(1)] = float(m.group(2))
for k, v in rows.items():
    p = sorted(x[1] for x in v)[len(v) // 2]; c = sorted(x[0] for x in v)[len(v) // 2]
    t = ms.get(k, float("nan"))
    print(f"{k:16s} sclk {c:5d} MHz  power {p:6.0f} W  launch {t * 1e3:8.1f} us  energy {p * t / 1e3:7.3f} J  cycles {c * t:9.0f} k")
PY
