#!/usr/bin/env python3
"""Developer A/B worker: the pooled conv2b activation (and conv3a's, one layer on) of the CURRENT kernel selection for a list of frame sizes -> npz.
SUPERSLAM_HIP_CONV2=fused|split (developer build, SSHIP_DEV_LIBRARY) forces conv_fuse2.hip / the two-launch path; tests/test_gpu_alt_paths.py
compares the two dumps bit for bit.  usage: fuse2_dump.py out.npz [HxWxB ...]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import _devlib; _devlib.use_dev_library()
from superslam_amd import SuperPoint, _lib
from superslam_amd.synth import make_frame
from superslam_amd.weights import make_superpoint_weights, save_safetensors

out = sys.argv[1]
sizes = sys.argv[2:] or ["200x328x2", "240x320x2", "376x1241x2", "96x250x4", "370x150x2", "64x96x2", "72x64x3", "376x1376x4", "1080x1920x2", "32x40x5"]
torch.cuda.set_device(0); _lib.init(0)
d = tempfile.mkdtemp(); save_safetensors(make_superpoint_weights(0), d + "/sp.safetensors")
res = {}
for sz in sizes:
    H, W, B = (int(v) for v in sz.split("x"))
    sp = SuperPoint(d + "/sp.safetensors", 300, 0.005, 4, max_batch=B); assert sp.initialize(), sp.last_error
    imgs = torch.from_numpy(np.stack([make_frame(H, W, 100 + i) for i in range(B)])).cuda()
    sp.extract_batch_device(imgs); torch.cuda.synchronize()
    for layer, (h, w, c) in ((3, (H // 4, W // 4, 64)), (4, (H // 4, W // 4, 128))):
        a = np.zeros((B, h, w, c), np.float16)
        _lib.check(_lib.lib().sship_sp_debug_activation(sp._h, layer, a.ctypes.data, a.nbytes))
        res[f"{sz}_L{layer}"] = a.view(np.uint16)
    sp.close()
np.savez(out, **res)
print("dumped", len(res), "arrays")
