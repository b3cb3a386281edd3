#!/usr/bin/env python3
"""Loop ONE stage for ~8 s so that rocm-smi can sample power / clocks: stage = conv1ab | conv2a | attn | ffn | nms | idle"""
import ctypes as C, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..')); import _devlib; _devlib.use_dev_library()  # SSHIP_DEV_LIBRARY -> explicit set_library_path (A/B builds)
from superslam_amd import LightGlue, SuperPoint, _lib
from superslam_amd.synth import make_stereo_pair
from superslam_amd.weights import make_lightglue_weights, make_superpoint_weights, save_safetensors
stage = sys.argv[1]; P = 64; H, W, K = 376, 1376, 600
torch.cuda.set_device(0); _lib.init(0); L = _lib.lib()
d = tempfile.mkdtemp()
save_safetensors(make_superpoint_weights(0), d + "/sp.safetensors"); save_safetensors(make_lightglue_weights(1), d + "/lg.safetensors")
sp = SuperPoint(d + "/sp.safetensors", K, 0.005, 4, max_batch=2 * P); assert sp.initialize()
lg = LightGlue(d + "/lg.safetensors", W, H, max_keypoints=K, max_pairs=P); assert lg.initialize()
l, r = make_stereo_pair(H, W, 1234)
imgs = torch.from_numpy(np.stack([l, r] * P)).cuda()
imgs = torch.stack([torch.roll(imgs[i], i * 7, 0) for i in range(2 * P)])
L.sship_set_profiling(1); desc, kp, n = sp.extract_batch_device(imgs); torch.cuda.synchronize(); L.sship_set_profiling(0)
lg.match_batch_device(kp, n, desc); torch.cuda.synchronize()
ms = C.c_float(0)
t0 = time.time(); it = 0
while time.time() - t0 < 9.0:
    if stage == "idle": time.sleep(0.2)
    elif stage in ("conv1ab", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "convPa", "convPb", "nms"): _lib.check(L.sship_sp_bench_layer(sp._h, {"conv1ab": 1, "conv2a": 2, "conv2b": 3, "conv3a": 4, "conv3b": 5, "conv4a": 6, "convPa": 8, "convPb": 9, "nms": 12}[stage], 2 * P, H, W, 50, C.byref(ms), None))
    else: _lib.check(L.sship_lg_bench_stage(lg._h, {"attn": 1, "xattn": 2, "ffn": 3, "ffnc": 4}[stage], 200, C.byref(ms)))
    it += 1
print(stage, "avg_ms", ms.value, "iters", it)
