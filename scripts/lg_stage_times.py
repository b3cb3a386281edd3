#!/usr/bin/env python3
"""Times the LightGlue stages (sship_lg_bench_stage) for a 64-pair batch of 600 keypoints: one line per stage.
usage: [SSHIP_DEV_LIBRARY=variant.so] python scripts/lg_stage_times.py [pairs] [max_kp]"""
import ctypes as C
import os
import sys
import tempfile

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '.')); import _devlib; _devlib.use_dev_library()  # SSHIP_DEV_LIBRARY -> explicit set_library_path (A/B builds)
from superslam_amd import LightGlue, _lib  # noqa: E402
from superslam_amd.weights import make_lightglue_weights, save_safetensors  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 600
_lib.init()
d = tempfile.mkdtemp()
save_safetensors(make_lightglue_weights(1), os.path.join(d, "lg.safetensors"))
lg = LightGlue(os.path.join(d, "lg.safetensors"), 1376, 376, max_keypoints=K, max_pairs=P)
assert lg.initialize(), lg.last_error
g = torch.Generator().manual_seed(0)
kp = torch.rand((2 * P, K, 3), generator=g) * torch.tensor([1376.0, 376.0, 1.0])
ds = torch.nn.functional.normalize(torch.randn((2 * P, K, 256), generator=g), dim=-1).half()
n = torch.full((2 * P,), K, dtype=torch.int32)
lg.match_batch_device(kp.cuda(), n.cuda(), ds.cuda())
torch.cuda.synchronize()
names = ["wqkv0", "self_attn", "cross_attn", "self_ffn", "cross_ffn", "last_ffn", "assign_lse", "assign_arg"]
out = []
for sid, name in enumerate(names):
    ms = C.c_float(0)
    _lib.check(_lib.lib().sship_lg_bench_stage(lg._h, sid, 20, C.byref(ms)))
    out.append(f"{name}={ms.value * 1e3:.1f}us")
print(os.environ.get("SSHIP_DEV_LIBRARY", "default").split("/")[-1], " ".join(out), flush=True)
