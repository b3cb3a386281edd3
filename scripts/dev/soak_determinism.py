#!/usr/bin/env python3
"""Soak: the same 64-pair front-end call N times on the same input - every output (keypoints, counts, descriptors, matches, scores) must equal the first call's
bit for bit (a pipeline-synchronisation bug in a persistent kernel shows up as a rare mismatch long before it shows up as a wrong test).
usage: python scripts/dev/soak_determinism.py [calls] [pairs]"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from superslam_amd import FrontEndBatch, LightGlue, SuperPoint, _lib
from superslam_amd.synth import make_stereo_pair
from superslam_amd.weights import make_lightglue_weights, make_superpoint_weights, save_safetensors

N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
P = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H, W, K = 376, 1376, 600
_lib.init(0)
d = tempfile.mkdtemp()
save_safetensors(make_superpoint_weights(0), d + "/sp.safetensors"); save_safetensors(make_lightglue_weights(1), d + "/lg.safetensors")
sp = SuperPoint(d + "/sp.safetensors", K, 0.005, 4, max_batch=2 * P); assert sp.initialize(), sp.last_error
lg = LightGlue(d + "/lg.safetensors", W, H, max_keypoints=K, max_pairs=P); assert lg.initialize(), lg.last_error
fe = FrontEndBatch(sp, lg, P, H, W)
pairs = [make_stereo_pair(H, W, 900 + i) for i in range(min(P, 8))]
x = torch.from_numpy(np.stack([im for i in range(P) for im in pairs[i % len(pairs)]])).cuda()
fe.run(x); torch.cuda.synchronize()
ref = [t.clone() for t in (fe.kp, fe.n, fe.desc, fe.matches0, fe.mscores0)]
bad = 0
for it in range(N):
    fe.run(x); torch.cuda.synchronize()
    for name, a, b in zip(("kp", "n", "desc", "matches0", "mscores0"), (fe.kp, fe.n, fe.desc, fe.matches0, fe.mscores0), ref):
        if not torch.equal(a, b):
            bad += 1
            print(f"call {it}: {name} differs in {int((a != b).sum())} elements", flush=True)
print(f"soak: {N} calls of {P} pairs, mismatching outputs: {bad}")
sys.exit(1 if bad else 0)
