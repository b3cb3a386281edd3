#!/usr/bin/env python3
"""Condense scripts/pmc_sq.sh output: one line per kernel with the wave-cycle split and MFMA-busy share.
usage: pmc_sq_table.py pmc_sq.txt"""
import re, sys
rows = {}
for ln in open(sys.argv[1]):
    m = re.match(r"(.{90}) (\S+)\s+n=\s*(\d+) mean=\s*([\d.]+)", ln)
    if m:
        rows.setdefault(m.group(1).strip(), {})[m.group(2)] = (int(m.group(3)), float(m.group(4)))
print(f"{'kernel':70s} {'n':>4s} {'wait_any':>8s} {'wait_inst':>9s} {'active':>7s} {'mfma_busy/busy':>14s} {'lds_conf/idx':>12s}")
for k, c in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", (0, 0))[1] * kv[1].get("SQ_WAVE_CYCLES", (0, 0))[0]):
    wc = c.get("SQ_WAVE_CYCLES", (0, 0))[1]
    if not wc:
        continue
    g = lambda n: c.get(n, (0, 0.0))[1]
    busy = g("SQ_BUSY_CYCLES")
    # BUSY_CYCLES is summed over the shader engines' SQs, MFMA_BUSY over SIMDs; report the raw ratio (relative between kernels)
    print(f"{k[:70]:70s} {c['SQ_WAVE_CYCLES'][0]:4d} {g('SQ_WAIT_ANY') / wc:8.2f} {g('SQ_WAIT_INST_ANY') / wc:9.2f} "
          f"{g('SQ_ACTIVE_INST_ANY') / wc:7.2f} {(g('SQ_VALU_MFMA_BUSY_CYCLES') / busy if busy else 0):14.3f} "
          f"{(g('SQ_LDS_BANK_CONFLICT') / g('SQ_LDS_IDX_ACTIVE') if g('SQ_LDS_IDX_ACTIVE') else 0):12.3f}")
