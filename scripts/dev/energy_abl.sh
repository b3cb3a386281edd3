#!/bin/bash
# Energy / time ablation of one kernel (developer aid; one script instead of the per-kernel copies of round 4):
#   scripts/dev/energy_abl.sh ffn  1 2 4 8 5 3     k_lg_ffn4,                 variants lib/variants/ffn4abl<n>.so  (build.py --variant ffn4abl<n> -DSSHIP_FFN4_ABL=<n>)
#   scripts/dev/energy_abl.sh attn 1 2 3           k_lg_attention,            variants attnabl<n>.so               (-DSSHIP_ATTN_ABL=<n>)
#   scripts/dev/energy_abl.sh conv1ab 1 2 3        conv1a + conv1b + pool,    variants ppabl<n>.so                 (-DSSHIP_PP_ABL=<n>)      (also: conv2a)
# Each variant loops the kernel for ~9 s (scripts/dev/loop_kernel.py <stage>) under rocm-smi (scripts/dev/power_poll.sh): median clock x
# power x launch time -> energy and shader cycles per launch.  The variants' results are wrong by design.  Output: gpurun_out/energy_abl_<stage>.txt
set -u
ST=$1; shift
case $ST in ffn) PFX=ffn4abl;; attn) PFX=attnabl;; *) PFX=ppabl;; esac
mkdir -p gpurun_out
O=gpurun_out/energy_abl_$ST.txt
: > $O
run() { tag=$1; shift; bash scripts/dev/power_poll.sh $tag "$@" >> $O 2>&1; tail -1 /tmp/pp_$tag.log | sed "s/^/$tag /" >> $O; }
run base python scripts/dev/loop_kernel.py $ST
for v in "$@"; do
  SSHIP_DEV_LIBRARY=$(pwd)/superslam_amd/lib/variants/$PFX$v.so run abl$v python scripts/dev/loop_kernel.py $ST
done
python - $O <<'PY'
import re, sys
txt = open(sys.argv[1]).read()
rows, ms = {}, {}
for line in txt.splitlines():
    m = re.match(r"^([a-zA-Z0-9_]+): .*sclk clock level: 1: \((\d+)Mhz\).*Power \(W\): ([0-9.]+)", line)
    if m: rows.setdefault(m.group(1), []).append((int(m.group(2)), float(m.group(3))))
    m = re.match(r"^([a-zA-Z0-9_]+) \w+ avg_ms ([0-9.]+)", line)
    if m: ms[m.group(1)] = float(m.group(2))
for k, v in rows.items():
    p = sorted(x[1] for x in v)[len(v) // 2]; c = sorted(x[0] for x in v)[len(v) // 2]
    t = ms.get(k, float("nan"))
    print(f"{k:10s} sclk {c:5d} MHz  power {p:6.0f} W  launch {t * 1e3:8.1f} us  energy {p * t:8.2f} mJ  cycles {c * t:9.0f} k")
PY
