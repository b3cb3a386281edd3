#!/usr/bin/env python3
"""What would Winograd F(2x2, 3x3) for conv1b cost in accuracy?  (CPU experiment with the oracle: analysis tool, not product.)

conv1b (64 -> 64 channels at full resolution) is 36 % of the pair's FLOPs and the dominant kernel; F(2x2, 3x3) needs 16 instead of
36 multiplies per output and channel pair (2.25x fewer MFMAs).  On an fp16 engine the transformed input tile V = B^T d B and the
transformed weights U = G g G^T are rounded to fp16 before the matrix multiply.  This script runs the SuperPoint oracle three ways on
synthetic 376 x 1376 frames and compares the extracted keypoints:
  A  fp32 everywhere (the reference definition)
  B  fp16 weights / activations, direct convolutions (what the HIP path does today)
  C  as B, but conv1b through Winograd with U and V rounded to fp16 (V computed in fp32 from fp16 activations, rounded once)
usage: python scripts/analysis/winograd_conv1b_accuracy.py [frames]"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import hostpath as H  # noqa: E402
from oracle import superpoint_ref as SR  # noqa: E402
from superslam_amd.synth import make_stereo_pair  # noqa: E402
from superslam_amd.weights import make_superpoint_weights  # noqa: E402

BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


def q(t):
    return t.half().float()


def winograd_conv3x3(x, w, b):
    """x [B,C,H,W] (fp16-representable), w [K,C,3,3] -> [B,K,H,W]; U, V rounded to fp16, fp32 accumulation and output transform."""
    Bn, C, Hh, Ww = x.shape
    K = w.shape[0]
    U = q(torch.einsum("ij,kcjl,ml->kcim", G, q(w), G))                      # [K,C,4,4]
    xp = F.pad(x, (1, 1, 1, 1))
    t = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                   # [B,C,Th,Tw,4,4]
    V = q(torch.einsum("ij,bcthjl,ml->bcthim", BT, t, BT))                   # [B,C,Th,Tw,4,4]
    M = torch.einsum("kcim,bcthim->bkthim", U, V)                            # fp32 accumulate over c
    Y = torch.einsum("ij,bkthjl,ml->bkthim", AT, M, AT)                      # [B,K,Th,Tw,2,2]
    Th, Tw = Y.shape[2], Y.shape[3]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(Bn, K, Th * 2, Tw * 2)
    return y + b.view(1, -1, 1, 1)


def encode(sd, image, mode):
    fp16 = mode in ("B", "C")
    qq = q if fp16 else (lambda t: t)
    x = qq(image)
    for name in SR.ENC:
        if mode == "C" and name == "conv1b":
            y = winograd_conv3x3(x, sd[name + ".weight"], sd[name + ".bias"])
        else:
            y = F.conv2d(x, qq(sd[name + ".weight"]), sd[name + ".bias"], padding=1)
        x = qq(F.relu(y))
        if name in SR.POOL_AFTER:
            x = F.max_pool2d(x, 2, 2)
    return x


def keypoints(sd, img_u8, mode, max_kp=600, thr=0.005):
    image = SR.preprocess_u8(torch.from_numpy(img_u8)[None])
    with torch.no_grad():
        feat = encode(sd, image, mode)
        logits = SR.detector_logits(sd, feat, mode != "A")
        scores = SR.nms(SR.heatmap_from_logits(logits), 4)[0].numpy()
        desc = SR.descriptor_grid(sd, feat, mode != "A")[0]
    h, w = img_u8.shape
    sel = H.select_topk(scores, h, w, thr, 4, max_kp, h // 8, w // 8)   # border 4 as the reference's config
    return sel["hw"], logits[0].numpy(), desc.numpy()


def iou(a, b):
    sa, sb = {tuple(p) for p in a}, {tuple(p) for p in b}
    return len(sa & sb) / max(1, len(sa | sb))


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    sd = make_superpoint_weights(0)
    rows = []
    for f in range(frames):
        left, right = make_stereo_pair(376, 1376, 4000 + f)
        for img in (left, right):
            out = {m: keypoints(sd, img, m) for m in "ABC"}
            kp = {m: out[m][0] for m in "ABC"}
            rows.append((iou(kp["B"], kp["A"]), iou(kp["C"], kp["A"]), iou(kp["C"], kp["B"]),
                         float(np.abs(out["B"][1] - out["A"][1]).max()), float(np.abs(out["C"][1] - out["A"][1]).max()),
                         float(np.abs(out["B"][2] - out["A"][2]).max()), float(np.abs(out["C"][2] - out["A"][2]).max())))
            print("IoU(B,A) %.4f  IoU(C,A) %.4f  IoU(C,B) %.4f | logits max|d| B %.3e  C %.3e | descriptor grid max|d| B %.3e  C %.3e" % rows[-1], flush=True)
    r = np.array(rows)
    print("mean: IoU(B,A) %.4f  IoU(C,A) %.4f  IoU(C,B) %.4f | min: %.4f %.4f %.4f" % (*r[:, :3].mean(0), *r[:, :3].min(0)))


if __name__ == "__main__":
    main()
