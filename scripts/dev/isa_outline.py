#!/usr/bin/env python3
"""Developer aid: outline of one kernel's ISA from a kstats.sh assembly file - runs of loads / MFMAs / LDS ops / stores with every
s_waitcnt, s_barrier and branch in between.   usage: isa_outline.py /tmp/kstats/lg_kernels.s <substring of the mangled name>"""
import sys
lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[2] in l.split(":")[0] and l.rstrip().split(";")[0].strip().endswith(":"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
keys = ("s_barrier", "s_waitcnt", "global_load", "buffer_load", "v_mfma", "global_store", "buffer_store", "s_cbranch", "ds_write", "ds_read", "scratch_", "v_accvgpr", "s_sleep")
prev, cnt, out = None, 0, []
def flush():
    global prev, cnt
    if prev: out.append(f"    {prev} x{cnt}")
    prev, cnt = None, 0
for l in lines[start + 1:end]:
    t = l.strip().split()[0] if l.strip() else ""
    kind = next((k for k in keys if t.startswith(k)), None)
    if kind is None:
        if t.endswith(":") and not t.startswith(";"): flush(); out.append(t)
        continue
    if kind in ("s_waitcnt", "s_barrier", "s_cbranch"):
        flush(); out.append("  " + " ".join(l.split()))
    elif kind == prev: cnt += 1
    else: flush(); prev, cnt = kind, 1
flush()
print("\n".join(out))
