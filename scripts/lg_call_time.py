#!/usr/bin/env python3
"""Times whole LightGlue calls (sship_lg_match_batch_device) on random descriptors: ms per call of P pairs.
usage: [SUPERSLAM_HIP_LG_SPLIT=1] python scripts/lg_call_time.py [pairs] [max_kp] [calls]"""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '.')); import _devlib; _devlib.use_dev_library()  # SSHIP_DEV_LIBRARY -> explicit set_library_path (A/B builds)
from superslam_amd import LightGlue, _lib  # noqa: E402
from superslam_amd.weights import make_lightglue_weights, save_safetensors  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 64
K = int(sys.argv[2]) if len(sys.argv) > 2 else 600
CALLS = int(sys.argv[3]) if len(sys.argv) > 3 else 20
_lib.init()
d = tempfile.mkdtemp()
save_safetensors(make_lightglue_weights(1), os.path.join(d, "lg.safetensors"))
lg = LightGlue(os.path.join(d, "lg.safetensors"), 1376, 376, max_keypoints=K, max_pairs=P)
assert lg.initialize(), lg.last_error
g = torch.Generator().manual_seed(0)
kp = (torch.rand((2 * P, K, 3), generator=g) * torch.tensor([1376.0, 376.0, 1.0])).cuda()
ds = torch.nn.functional.normalize(torch.randn((2 * P, K, 256), generator=g), dim=-1).half().cuda()
n = torch.full((2 * P,), K, dtype=torch.int32).cuda()
for _ in range(3):
    lg.match_batch_device(kp, n, ds)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(CALLS):
    lg.match_batch_device(kp, n, ds)
e1.record()
torch.cuda.synchronize()
print(f"split={os.environ.get('SUPERSLAM_HIP_LG_SPLIT', 'default')} pairs={P} kp={K}: {e0.elapsed_time(e1) / CALLS:.3f} ms per call", flush=True)
