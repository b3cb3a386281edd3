#!/usr/bin/env python3
"""Times every SuperPoint layer (sship_sp_bench_layer) at the headline shape: one line.  usage: [SSHIP_DEV_LIBRARY=...] python scripts/sp_layer_times.py [pairs]"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '.')); import _devlib; _devlib.use_dev_library()  # SSHIP_DEV_LIBRARY -> explicit set_library_path (A/B builds)
from superslam_amd import SuperPoint, _lib  # noqa: E402
from superslam_amd.synth import make_stereo_pair  # noqa: E402
from superslam_amd.weights import make_superpoint_weights, save_safetensors  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
H, W = 376, 1376
_lib.init()
d = tempfile.mkdtemp()
save_safetensors(make_superpoint_weights(0), os.path.join(d, "sp.safetensors"))
sp = SuperPoint(os.path.join(d, "sp.safetensors"), 600, 0.005, 4, max_batch=2 * P)
assert sp.initialize(), sp.last_error
l, r = make_stereo_pair(H, W, 1234)
imgs = torch.from_numpy(np.stack([l, r] * P)).cuda()
imgs = torch.stack([torch.roll(imgs[i], i * 7, 0) for i in range(2 * P)])
_lib.lib().sship_set_profiling(1)  # the handle keeps a copy of this call's pixels: the layers are re-launched on real data
sp.extract_batch_device(imgs)
torch.cuda.synchronize()
_lib.lib().sship_set_profiling(0)
names = ["c1a*", "conv1ab", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convPb", "cDa*", "cDb*", "nms", "topk", "desc"]
out = []
for lid, name in enumerate(names):
    if name.endswith("*"):
        continue
    ms = C.c_float(0)
    _lib.check(_lib.lib().sship_sp_bench_layer(sp._h, lid, 2 * P, H, W, 10, C.byref(ms), None))
    out.append(f"{name}={ms.value * 1e3:.0f}")
print(os.environ.get("SSHIP_DEV_LIBRARY", "default").split("/")[-1], " ".join(out), "(us)", flush=True)
