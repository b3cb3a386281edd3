mkdir -p gpurun_out
O=gpurun_out/r04_p2_power.txt
: > $O
run() { tag=$1; shift; bash scripts/dev/power_poll.sh $tag "$@" >> $O 2>&1; tail -1 /tmp/pp_$tag.log >> $O; }
SUPERSLAM_HIP_FFN=16 run ffn16 python scripts/dev/loop_kernel.py ffn
SUPERSLAM_HIP_CONV64=wino run wino2a python scripts/dev/loop_kernel.py conv2a
run conv2b python scripts/dev/loop_kernel.py conv2b
run conv3b python scripts/dev/loop_kernel.py conv3b
run conv4a python scripts/dev/loop_kernel.py conv4a
run convPa python scripts/dev/loop_kernel.py convPa
run convPb python scripts/dev/loop_kernel.py convPb
python - <<'PY'
import re
txt = open("gpurun_out/r04_p2_power.txt").read()
rows = {}
for line in txt.splitlines():
    m = re.match(r"^([a-zA-Z0-9]+): .*sclk clock level: 1: \((\d+)Mhz\).*Power \(W\): ([0-9.]+)", line)
    if m: rows.setdefault(m.group(1), []).append((int(m.group(2)), float(m.group(3))))
for k, v in rows.items():
    t = [l for l in txt.splitlines() if "avg_ms" in l]
    print(f"{k:8s} sclk {min(x[0] for x in v)}..{max(x[0] for x in v)} MHz  power {min(x[1] for x in v):.0f}..{max(x[1] for x in v):.0f} W")
print("\n".join(l for l in txt.splitlines() if "avg_ms" in l))
PY
