#!/bin/bash
# Developer aid: per-kernel registers / spills / code size of one .hip source, device-only compile (no GPU needed).
# usage: scripts/dev/kstats.sh superslam_amd/csrc/lg_ffn16.hip [extra hipcc flags]   ->  /tmp/kstats/<name>.s + a summary
set -e
src=$1; shift
name=$(basename "$src" .hip)
mkdir -p /tmp/kstats
flags="-fno-honor-nans"
case "$name" in lg_*|conv_wino) flags="$flags -Xclang -target-feature -Xclang -packed-fp32-ops";; esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $flags "$@" --cuda-device-only -S "$src" -o /tmp/kstats/$name.s
python3 - /tmp/kstats/$name.s <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
for m in re.finditer(r"\.set (_Z\S+)\.has_indirect_call, \d+\n\t\.section\t\.AMDGPU\.csdata.*?\n; Kernel info:\n(.*?); WaveLimiterHint", txt, re.S):
    nm, blk = m.group(1), m.group(2)
    g = lambda k: int(re.search(r"; %s: (\d+)" % k, blk).group(1))
    code = int(re.search(r"; codeLenInByte = (\d+)", blk).group(1))
    dem = subprocess.run(["c++filt", nm], capture_output=True, text=True).stdout.strip().split("(")[0][:78]
    print(f"{dem:80s} code {code:7d} B  vgpr {g('NumVgprs'):3d} agpr {g('NumAgprs'):3d} total {g('TotalNumVgprs'):3d} sgpr {g('TotalNumSgprs'):3d} scratch {g('ScratchSize'):4d} occ {g('Occupancy')}")
PY
