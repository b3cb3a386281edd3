#!/usr/bin/env python3
"""conv2a / conv2b of the current kernel selection against a CPU convolution of the previous layer's activation (fp16 inputs, fp32 math)."""
import os, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.nn.functional as F
import os as _os, sys as _sys; _sys.path.insert(0, _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), '..')); import _devlib; _devlib.use_dev_library()  # SSHIP_DEV_LIBRARY -> explicit set_library_path (A/B builds)
from superslam_amd import SuperPoint, _lib
from superslam_amd.synth import make_stereo_pair
from superslam_amd.weights import make_superpoint_weights, save_safetensors

H, W = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "64x96").split("x"))
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
torch.cuda.set_device(0); _lib.init(0)
d = tempfile.mkdtemp(); spw = make_superpoint_weights(0)
if os.environ.get("WINO_IDENTITY"):
    for nm in ("conv2a", "conv2b"):
        w = torch.zeros((64, 64, 3, 3)); w[torch.arange(64), torch.arange(64), 1, 1] = 1.0
        spw[nm + ".weight"] = w; spw[nm + ".bias"] = torch.zeros(64)
save_safetensors(spw, d + "/sp.safetensors")
sp = SuperPoint(d + "/sp.safetensors", 200, 0.005, 4, max_batch=B); assert sp.initialize()
l, r = make_stereo_pair(H, W, 5)
imgs = torch.from_numpy(np.stack([l, r] * (B // 2))).cuda()
sp.extract_batch_device(imgs); torch.cuda.synchronize()
def act(layer, h, w, c):
    a = np.zeros((B, h, w, c), np.float16)
    _lib.check(_lib.lib().sship_sp_debug_activation(sp._h, layer, a.ctypes.data, a.nbytes))
    return torch.from_numpy(a.astype(np.float32)).permute(0, 3, 1, 2)
a1b = act(1, H // 2, W // 2, 64); a2a = act(2, H // 2, W // 2, 64); a2b = act(3, H // 4, W // 4, 64)
ref2a = F.relu(F.conv2d(a1b, spw["conv2a.weight"].half().float(), spw["conv2a.bias"], padding=1))
ref2b = F.max_pool2d(F.relu(F.conv2d(a2a, spw["conv2b.weight"].half().float(), spw["conv2b.bias"], padding=1)), 2, 2)
if os.environ.get("WINO_PATTERN"):
    # debug build -DSSHIP_WINO_DBG=1: channel 32 m + 8 g + 4 hh + e of pixel (i, jj) of its 2x2 tile holds 16 (2 i + jj) + 4 g + e
    ch = torch.arange(64); r = 4 * ((ch % 32) // 8) + (ch % 4)
    yy, xx = torch.meshgrid(torch.arange(H // 2), torch.arange(W // 2), indexing="ij")
    exp = (16 * (2 * (yy % 2) + (xx % 2)))[None, None] + r[None, :, None, None]
    bad = torch.nonzero(a2a != exp.float())
    print("pattern mismatches", len(bad), "of", a2a.numel(), "first", bad[:8].tolist(), "x hist", torch.bincount(bad[:, 3], minlength=W // 2).tolist() if len(bad) else [],
          "ch hist", torch.bincount(bad[:, 1], minlength=64).tolist() if len(bad) else [])
    sys.exit(0)
for name, got, ref in (("conv2a", a2a, ref2a), ("conv2b+pool", a2b, ref2b)):
    nf = ~torch.isfinite(got)
    if nf.any():
        idx = torch.nonzero(nf)
        print(f"   {name}: non-finite {int(nf.sum())} of {nf.numel()}; y hist", torch.bincount(idx[:, 2], minlength=got.shape[2]).tolist(), "x hist",
              torch.bincount(idx[:, 3], minlength=got.shape[3]).tolist(), "ch hist", torch.bincount(idx[:, 1], minlength=64).tolist())
        got = torch.where(nf, torch.zeros_like(got), got)
    dd = (got - ref).abs()
    print(f"{os.environ.get('SUPERSLAM_HIP_CONV64', 'direct'):6s} {name}: max|d| {float(dd.max()):.4e} mean|d| {float(dd.mean()):.3e} ref max {float(ref.abs().max()):.3f} "
          f"finite {bool(torch.isfinite(got).all())}")
    if float(dd.max()) > 0.05:
        idx = torch.nonzero(dd > 0.05)
        print("   mismatches", len(idx), "first", idx[:6].tolist(), "rows(y) hist", torch.bincount(idx[:, 2], minlength=got.shape[2])[:24].tolist(),
              "cols(x) hist", torch.bincount(idx[:, 3], minlength=got.shape[3])[:40].tolist(), "ch hist", torch.bincount(idx[:, 1], minlength=64).tolist())
        b0, c0, y0, x0 = idx[0].tolist()
        print("   got", got[b0, c0, y0, max(0, x0 - 2):x0 + 3].tolist(), "ref", ref[b0, c0, y0, max(0, x0 - 2):x0 + 3].tolist())
        if os.environ.get("WINO_IDENTITY"):
            np.set_printoptions(precision=3, linewidth=250, suppress=True)
            for (yy, xx) in ((y0, x0), (y0 + 1, x0), (y0, x0 + 1), (y0 + 5, x0 + 3)):
                g_, r_ = got[b0, :, yy, xx].numpy(), ref[b0, :, yy, xx].numpy()
                bad = np.nonzero(np.abs(g_ - r_) > 0.05)[0]
                print(f"   pixel (y {yy}, x {xx}): bad channels {bad.tolist()} got {g_[bad]} ref {r_[bad]}")
                for ch in bad[:4]:   # where does the wrong value come from?
                    hit = torch.nonzero((ref[b0] - float(g_[ch])).abs() < 1e-4)
                    print(f"      ch {ch}: got {g_[ch]:.4f} appears in ref at (ch, y, x) {hit[:6].tolist()}")
sp.close()
