#!/usr/bin/env python3
"""EigenPlaces timing on the GPU box: per-descriptor latency of the synchronous host entry points (median of `n` calls, wall clock) and of
the device-resident path (sship_ep_bench: back-to-back launches on the handle's stream, device events).
usage: python scripts/ep_time.py [n] [--loop-only]   (--loop-only: just the device loop, for rocprofv3 --kernel-trace --stats)"""
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import _devlib  # noqa: E402

_devlib.use_dev_library()   # A/B only: SSHIP_DEV_LIBRARY names the developer build (SUPERSLAM_HIP_EP_STEM=gemm); unset = the shipped library
from superslam_amd import _lib  # noqa: E402
from superslam_amd import eigenplaces as P  # noqa: E402
from superslam_amd.synth import make_frame  # noqa: E402
from superslam_amd.weights import make_eigenplaces_weights, save_safetensors  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 50
loop_only = "--loop-only" in sys.argv
_lib.init(0)
L = _lib.lib()
d = tempfile.mkdtemp()
path = os.path.join(d, "ep.safetensors")
save_safetensors(make_eigenplaces_weights(2), path)
h = C.c_void_p()
_lib.check(L.sship_ep_create(path.encode(), 512, 512, C.byref(h)))
img = make_frame(376, 1241, 3)
dimg = torch.from_numpy(img).cuda()
out = np.zeros(512, np.float32)
ms = C.c_float(0)
_lib.check(L.sship_ep_bench(h, dimg.data_ptr(), 376, 1241, 1241, 1, n, C.byref(ms)))
res = {"device_loop_ms": round(ms.value, 4), "iters": n, "image": [376, 1241], "input": [512, 512]}
if not loop_only:
    def med(fn):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        return round(ts[len(ts) // 2], 4)

    res["host_u8_sync_ms"] = med(lambda: _lib.check(L.sship_ep_infer_u8(h, img.ctypes.data, 376, 1241, 1241, 1, out.ctypes.data)))
    x = P.preprocess(img, 512, 512)
    res["host_fp32_sync_ms"] = med(lambda: _lib.check(L.sship_ep_infer(h, x.ctypes.data, out.ctypes.data)))
    res["host_preprocess_ms"] = med(lambda: P.preprocess(img, 512, 512))
    res["gflop"] = 19.0
print(json.dumps(res), flush=True)
L.sship_ep_destroy(h)
