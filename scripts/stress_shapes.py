"""Run the fused front-end over unusual shapes / keypoint budgets and check invariants (no parity claim here:
the parity suite covers numerics; this catches crashes, out-of-range indices and NaNs).  usage: stress_shapes.py"""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '.')); import _devlib; _devlib.use_dev_library()  # SSHIP_DEV_LIBRARY -> explicit set_library_path (A/B builds)
from superslam_amd import FrontEndBatch, LightGlue, SuperPoint, _lib
from superslam_amd.synth import make_stereo_pair
from superslam_amd.weights import make_lightglue_weights, make_superpoint_weights, save_safetensors
_lib.init(0)
d = tempfile.mkdtemp()
save_safetensors(make_superpoint_weights(0), d + "/sp.safetensors"); save_safetensors(make_lightglue_weights(1), d + "/lg.safetensors")
cases = [(64, 64, 600, 1), (72, 136, 64, 3), (480, 752, 600, 5), (720, 1280, 2048, 2), (376, 1241, 4096, 1), (376, 1376, 600, 7), (200, 328, 1, 2)]
for (H, W, mk, P) in cases:
    sp = SuperPoint(d + "/sp.safetensors", mk, 0.005, 4, max_batch=2 * P); assert sp.initialize(), sp.last_error
    lg = LightGlue(d + "/lg.safetensors", W, H, max_keypoints=mk, max_pairs=P); assert lg.initialize(), lg.last_error
    pairs = [make_stereo_pair(H, W, 900 + i) for i in range(P)]
    imgs = torch.from_numpy(np.stack([im for p in pairs for im in p])).cuda()
    fe = FrontEndBatch(sp, lg, P, H, W)
    for _ in range(2):
        fe.run(imgs, 0)
    torch.cuda.synchronize(); _lib.lib().sship_device_synchronize()
    n = fe.n.cpu().numpy(); m = fe.matches0.cpu().numpy(); ms = fe.mscores0.cpu().numpy()
    kp = fe.kp.cpu().numpy(); desc = fe.desc.float().cpu().numpy()
    assert (n >= 0).all() and (n <= mk).all(), n
    for p in range(P):
        n0, n1 = int(n[2 * p]), int(n[2 * p + 1])
        mm = m[p, :n0]
        assert ((mm >= -1) & (mm < max(n1, 1))).all(), (H, W, mk, p)
        assert np.isfinite(ms[p, :n0]).all() and (ms[p, :n0] >= 0).all() and (ms[p, :n0] <= 1 + 1e-5).all()
        assert (m[p, n0:] == -1).all()
        for s_, nn in ((2 * p, n0), (2 * p + 1, n1)):
            if nn:
                k = kp[s_, :nn]
                assert (k[:, 0] >= 0).all() and (k[:, 0] <= W).all() and (k[:, 1] >= 0).all() and (k[:, 1] <= H).all()
                assert (np.diff(k[:, 2]) <= 0).all()
                nr = np.linalg.norm(desc[s_, :nn], axis=1)
                assert np.allclose(nr, 1.0, atol=3e-3), (nr.min(), nr.max())
    print(f"{H}x{W} max_kp={mk} pairs={P}: n={n.tolist()[:6]} matches={[int((m[p] >= 0).sum()) for p in range(min(P, 3))]} ok", flush=True)
    sp.close(); lg.close()
print("stress ok")
