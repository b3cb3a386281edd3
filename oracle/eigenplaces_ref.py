"""CPU restatement of the EigenPlaces(ResNet18, 512-d) place recogniser (TEST INFRASTRUCTURE).

PARITY: TRUNK PINNED AGAINST transformers' ResNet-18 (oracle/pin_hf.py: bit-identical feature maps in fp64 with the same
weights, tests/test_oracle_pins_hf.py); AGGREGATION HEAD PINNED BY FORMULA ONLY (tests/test_eigenplaces.py: a second numpy fp64 derivation + torch's
lp_pool2d agree to 1e-12, seven mutations break it) - the hub package itself is absent; PREPROCESSING (OpenCV cv::resize): coordinates, clamping
and weights pinned against torch's independent bilinear interpolation to < 1 gray level (tests/test_eigenplaces.py), the fixed-point rounding
sequence rests on OpenCV's published source (OpenCV absent).  The network comes from ``torch.hub.load("gmberton/eigenplaces", "get_trained_model", backbone="ResNet18",
fc_output_dim=512)`` (/root/reference/utils/convert_eigenplaces_to_onnx.py:54-60), i.e. third-party code + torchvision's
ResNet-18, neither present in /root/reference nor in this image, and the pre/post-processing uses OpenCV (absent).  This file
restates the published definitions:
  * backbone = torchvision ResNet-18 without avgpool / fc (eigenplaces_model.get_backbone: ``list(backbone.children())[:-2]``),
    state-dict keys ``backbone.{0,1,4..7}...`` because the trunk is an nn.Sequential;
  * aggregation = L2Norm -> GeM(p = 3 learnable, eps = 1e-6) -> Flatten -> Linear(512, 512) -> L2Norm;
  * host preprocessing, src/EigenPlaces.cc:123-145: gray -> RGB replicate / BGR -> RGB, cv::resize(INTER_LINEAR) to
    (input_w, input_h) on u8, x 1/255, ImageNet mean / std, HWC -> CHW fp32;
  * cv::resize 8-bit bilinear as published in OpenCV imgproc/resize.cpp: source coordinate (d + 0.5) * scale - 0.5, 11-bit
    fixed-point coefficients (saturate_cast<short>(w * 2048)), horizontal pass in int32, vertical pass
    ``(((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2``;
  * post-processing, :147-174 + src/PlaceRecognizer.cc: L2 normalisation (a no-op after the model's own), cosine index,
    temporal-consistency voter.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

MEAN = (0.485, 0.456, 0.406)
STD = (0.229, 0.224, 0.225)
LAYERS = ((4, 64, 1), (5, 128, 2), (6, 256, 2), (7, 512, 2))   # (index in the Sequential, planes, stride of block 0)


def resize_bilinear_u8(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """cv::resize(src, dst, Size(out_w, out_h), 0, 0, INTER_LINEAR) for CV_8UC{1,3} (fixed-point path)."""
    a = np.ascontiguousarray(img, np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape

    def coeffs(n_dst, n_src):
        scale = n_src / n_dst
        f = (np.arange(n_dst, dtype=np.float64) + 0.5) * scale - 0.5
        f = f.astype(np.float32)                      # OpenCV computes fx in float
        s = np.floor(f).astype(np.int64)
        fr = (f - s).astype(np.float32)
        lo = s < 0
        s[lo] = 0; fr[lo] = 0.0
        hi = s >= n_src - 1
        s[hi] = n_src - 1; fr[hi] = 0.0
        s1 = np.minimum(s + 1, n_src - 1)
        c1 = np.rint(fr * np.float32(2048.0)).astype(np.int64)    # saturate_cast<short>: round to nearest even
        c0 = np.rint((np.float32(1.0) - fr) * np.float32(2048.0)).astype(np.int64)
        return s, s1, c0, c1

    sx, sx1, ax0, ax1 = coeffs(out_w, w)
    sy, sy1, by0, by1 = coeffs(out_h, h)
    src = a.astype(np.int64)
    rows = src[:, sx] * ax0[None, :, None] + src[:, sx1] * ax1[None, :, None]       # [h, out_w, c] int (x 2^11)
    r0, r1 = rows[sy], rows[sy1]
    out = (((by0[:, None, None] * (r0 >> 4)) >> 16) + ((by1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8).reshape(out_h, out_w, c)


def preprocess(image: np.ndarray, input_w: int, input_h: int) -> np.ndarray:
    """src/EigenPlaces.cc:123-145 -> fp32 [3, H, W] (ImageNet-normalised RGB)."""
    img = np.ascontiguousarray(image, np.uint8)
    rgb = np.repeat(img[:, :, None], 3, 2) if img.ndim == 2 else img[:, :, ::-1]      # GRAY2RGB / BGR2RGB
    rgb = resize_bilinear_u8(rgb, input_h, input_w)
    x = rgb.astype(np.float32) * np.float32(1.0 / 255.0)
    x = (x - np.array(MEAN, np.float32)) / np.array(STD, np.float32)
    return np.ascontiguousarray(x.transpose(2, 0, 1))


def _bn(sd, p, x):
    return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)


def _block(sd, p, x, stride):
    y = F.relu(_bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, stride, 1)))
    y = _bn(sd, p + ".bn2", F.conv2d(y, sd[p + ".conv2.weight"], None, 1, 1))
    if (p + ".downsample.0.weight") in sd:
        x = _bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0))
    return F.relu(y + x)


def backbone(sd, x):
    x = F.relu(_bn(sd, "backbone.1", F.conv2d(x, sd["backbone.0.weight"], None, 2, 3)))
    x = F.max_pool2d(x, 3, 2, 1)
    for idx, _planes, stride in LAYERS:
        x = _block(sd, f"backbone.{idx}.0", x, stride)
        x = _block(sd, f"backbone.{idx}.1", x, 1)
    return x


def aggregation(sd, feat):
    x = F.normalize(feat, p=2.0, dim=1)                                   # L2Norm over channels, per location
    p = sd["aggregation.1.p"]
    x = F.avg_pool2d(x.clamp(min=1e-6).pow(p), (x.size(-2), x.size(-1))).pow(1.0 / p)   # GeM
    x = x.flatten(1)
    x = F.linear(x, sd["aggregation.3.weight"], sd["aggregation.3.bias"])
    return F.normalize(x, p=2.0, dim=1)


def forward(sd, x, dtype=torch.float32, return_internals=False):
    """x: [B,3,H,W] preprocessed -> [B,512] L2-normalised descriptors."""
    sd = {k: v.to(dtype) for k, v in sd.items() if v.is_floating_point()}
    with torch.no_grad():
        feat = backbone(sd, x.to(dtype))
        out = aggregation(sd, feat)
    return (out, feat) if return_internals else out


def compute_global_descriptor(sd, image: np.ndarray, input_w=512, input_h=512, dtype=torch.float32) -> np.ndarray:
    x = torch.from_numpy(preprocess(image, input_w, input_h))[None]
    d = forward(sd, x, dtype)[0].float().numpy()
    n = np.linalg.norm(d)
    return d / n if n > 0 else d                                           # cv::normalize(desc, desc, 1.0, 0.0, NORM_L2)


class CosineDescriptorIndex:
    """src/PlaceRecognizer.cc:22-56."""

    def __init__(self):
        self.ids, self.db = [], []

    @staticmethod
    def _row(d):
        r = np.asarray(d, np.float32).reshape(-1)
        n = float(np.sqrt((r.astype(np.float64) ** 2).sum()))
        return (r / np.float32(n)) if n > 1e-12 else r

    def add(self, keyframe_id, desc):
        self.ids.append(int(keyframe_id)); self.db.append(self._row(desc))

    def query(self, desc, exclude_recent, top_k, min_score):
        m = len(self.ids)
        if m == 0 or m <= exclude_recent:
            return []
        q = self._row(desc)
        limit = m - exclude_recent
        scores = np.stack(self.db[:limit]) @ q
        out = [(self.ids[i], float(scores[i])) for i in range(limit) if scores[i] >= min_score]
        out.sort(key=lambda t: -t[1])
        return out[:top_k] if top_k > 0 else out


class TemporalConsistencyVoter:
    """src/PlaceRecognizer.cc:58-71."""

    def __init__(self, required_votes, id_tolerance):
        self.required, self.tol, self.streak, self.last, self.have = required_votes, id_tolerance, 0, 0, False

    def vote(self, best_id):
        if best_id is None:
            self.streak, self.have = 0, False
            return False
        consistent = self.have and abs(best_id - self.last) <= self.tol
        self.streak = self.streak + 1 if consistent else 1
        self.last, self.have = best_id, True
        return self.streak >= self.required
