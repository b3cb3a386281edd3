"""Experiment: the same 32 pairs as N concurrent sub-batches on N handle pairs (each handle owns its HIP stream).
usage: exp_streams.py [pairs] [steps]"""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '.')); import _devlib; _devlib.use_dev_library()  # SSHIP_DEV_LIBRARY -> explicit set_library_path (A/B builds)
from superslam_amd import FrontEndBatch, LightGlue, SuperPoint, _lib
from superslam_amd.synth import make_stereo_pair
from superslam_amd.weights import make_lightglue_weights, make_superpoint_weights, save_safetensors
P = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
H, W = 376, 1376
_lib.init(0)
d = tempfile.mkdtemp()
save_safetensors(make_superpoint_weights(0), d + "/sp.safetensors"); save_safetensors(make_lightglue_weights(1), d + "/lg.safetensors")
pairs = [make_stereo_pair(H, W, 1234 + i) for i in range(P)]
imgs = torch.from_numpy(np.stack([im for p in pairs for im in p])).cuda()
for ns in [int(x) for x in os.environ.get("NS", "1,2,4").split(",")]:
    sub = P // ns
    hs = []
    for k in range(ns):
        sp = SuperPoint(d + "/sp.safetensors", 600, 0.005, 4, max_batch=2 * sub); assert sp.initialize()
        lg = LightGlue(d + "/lg.safetensors", W, H, max_keypoints=600, max_pairs=sub); assert lg.initialize()
        hs.append((sp, lg, FrontEndBatch(sp, lg, sub, H, W), imgs[2 * sub * k: 2 * sub * (k + 1)]))
    def step():
        for sp, lg, fe, im in hs:
            fe.run(im, 0)
    for _ in range(3): step()
    torch.cuda.synchronize(); _lib.lib().sship_device_synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    _lib.lib().sship_device_synchronize(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"streams={ns} sub-batch={sub}: {P * steps / dt:.1f} pairs/s, {dt / steps * 1e3:.3f} ms/step", flush=True)
    for sp, lg, fe, im in hs:
        sp.close(); lg.close()
