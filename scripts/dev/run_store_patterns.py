#!/usr/bin/env python3
"""scripts/ubench/store_patterns under the power poller: time, GB/s and joules per launch for every (pattern, stride)."""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "scripts"))
import power_telemetry
exe = os.path.join(ROOT, "scripts", "ubench", "store_patterns")
rows = []
with power_telemetry.PowerPoller(0, hz=50.0) as p:
    for stride in (128, 256, 512):
        for pat in (0, 1, 2, 3):
            ta = time.monotonic()
            out = subprocess.run([exe, str(pat), str(stride), "1.5"], capture_output=True, text=True, timeout=120).stdout
            tb = time.monotonic()
            j = json.loads(out.strip().splitlines()[-1])
            w = p.window(tb - 1.2, tb - 0.1)      # the tail of the run: the timed loop
            j.update({"avg_W": w["avg_W"] if w else None, "sclk_MHz": w["sclk_MHz"] if w else None,
                      "joules_per_launch": round(w["avg_W"] * j["launch_us"] * 1e-6, 4) if w else None})
            rows.append(j); print(json.dumps(j), flush=True)
