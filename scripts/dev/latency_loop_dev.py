#!/usr/bin/env python3
"""Single-pair latency loop (the reference's per-frame unit): P = 1 through sship_frontend_batch_device and through the
synchronous host API (extract_stereo + match, what the reference-side adapters call).  Run under `rocprofv3 --kernel-trace --stats`
for the per-kernel timeline of one frame.  usage: python scripts/latency_loop.py [frames] [max_kp]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)) + "/.."); import _devlib; _devlib.use_dev_library()
from superslam_amd import FrontEndBatch, LightGlue, SuperPoint, _lib  # noqa: E402
from superslam_amd.synth import make_stereo_pair  # noqa: E402
from superslam_amd.weights import make_lightglue_weights, make_superpoint_weights, save_safetensors  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
K = int(sys.argv[2]) if len(sys.argv) > 2 else 600
H, W = 376, 1376
_lib.init()
d = tempfile.mkdtemp()
save_safetensors(make_superpoint_weights(0), os.path.join(d, "sp.safetensors"))
save_safetensors(make_lightglue_weights(1), os.path.join(d, "lg.safetensors"))
sp = SuperPoint(os.path.join(d, "sp.safetensors"), K, 0.005, 4, max_batch=2)
lg = LightGlue(os.path.join(d, "lg.safetensors"), W, H, max_keypoints=K, max_pairs=1)
assert sp.initialize() and lg.initialize()
l, r = make_stereo_pair(H, W, 1234)
x = torch.from_numpy(np.stack([l, r])).cuda()
fe = FrontEndBatch(sp, lg, 1, H, W)
for _ in range(5):
    fe.run(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    fe.run(x)
torch.cuda.synchronize()
dev_ms = (time.perf_counter() - t0) / N * 1e3
# one call at a time, synchronised: what a caller that needs the result before the next frame sees
t0 = time.perf_counter()
for _ in range(N):
    fe.run(x); torch.cuda.synchronize()
sync_ms = (time.perf_counter() - t0) / N * 1e3
for _ in range(3):
    fl, fr = sp.extract_stereo(l, r); lg.match(fl.keypoints, fl.descriptors, fr.keypoints, fr.descriptors)
t0 = time.perf_counter()
for _ in range(N):
    fl, fr = sp.extract_stereo(l, r)
    m = lg.match(fl.keypoints, fl.descriptors, fr.keypoints, fr.descriptors)
host_ms = (time.perf_counter() - t0) / N * 1e3
print(f"single pair {W}x{H}, {K} kp: device-resident back-to-back {dev_ms:.3f} ms, device-resident + sync per frame {sync_ms:.3f} ms, "
      f"host API (extract_stereo + match, numpy wrappers) {host_ms:.3f} ms; matches {len(m)}", flush=True)
sp.close(); lg.close()
