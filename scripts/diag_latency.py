#!/usr/bin/env python3
"""Diagnostic: P = 1 latency through handles sized like bench.py's (max_batch 128 / max_pairs 64) vs per-frame handles, fresh and
after a 64-pair call / a profiled call.  usage: python scripts/diag_latency.py"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superslam_amd import FrontEndBatch, LightGlue, SuperPoint, _lib  # noqa: E402
from superslam_amd.synth import make_stereo_pair  # noqa: E402
from superslam_amd.weights import make_lightglue_weights, make_superpoint_weights, save_safetensors  # noqa: E402

H, W, K = 376, 1376, 600
_lib.init()
L = _lib.lib()
d = tempfile.mkdtemp()
save_safetensors(make_superpoint_weights(0), os.path.join(d, "sp.safetensors"))
save_safetensors(make_lightglue_weights(1), os.path.join(d, "lg.safetensors"))
l, r = make_stereo_pair(H, W, 1234)
x2 = torch.from_numpy(np.stack([l, r])).cuda()
x128 = x2.repeat(64, 1, 1).contiguous()


def lat(fe, x, n=50, tag=""):
    for _ in range(5):
        fe.run(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fe.run(x)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    print(f"{tag}: {ms:.3f} ms per call", flush=True)
    return ms


for mb, mp in ((2, 1), (128, 64)):
    sp = SuperPoint(os.path.join(d, "sp.safetensors"), K, 0.005, 4, max_batch=mb)
    lg = LightGlue(os.path.join(d, "lg.safetensors"), W, H, max_keypoints=K, max_pairs=mp)
    assert sp.initialize() and lg.initialize()
    fe1 = FrontEndBatch(sp, lg, 1, H, W)
    lat(fe1, x2, tag=f"handles max_batch={mb} max_pairs={mp}: P=1 fresh")
    if mp == 64:
        fe64 = FrontEndBatch(sp, lg, 64, H, W)
        lat(fe64, x128, n=5, tag="  P=64")
        lat(fe1, x2, tag="  P=1 after P=64 calls")
        L.sship_set_profiling(2)
        fe64.run(x128); torch.cuda.synchronize(); _lib.stage_timings()
        L.sship_set_profiling(0)
        lat(fe1, x2, tag="  P=1 after a level-2 profiled P=64 call")
        L.sship_set_profiling(1)
        fe1.run(x2); torch.cuda.synchronize(); print("  stages P=1:", {k: round(v, 4) for k, v in _lib.stage_timings().items()})
        L.sship_set_profiling(0)
    sp.close(); lg.close()
