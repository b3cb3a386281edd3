"""CPU restatement of the SuperPoint dense network (TEST INFRASTRUCTURE - see oracle/__init__.py).

Follows /root/reference/utils/convert_superpoint_to_onnx.py:
  * encoder conv1a..conv4b, 3x3 s1 p1 + ReLU, MaxPool2d(2,2) after 1b/2b/3b ........ :38-45, :51-64
  * detector head convPa(3x3)+ReLU, convPb(1x1) -> softmax(dim=1) -> drop dustbin ... :46-47, :77-78
  * depth-to-space: pixel (8h+dy, 8w+dx) <- channel 8*dy+dx ......................... :79-81
  * NMS: max_pool2d(9, stride 1, pad 4); s = (s == pooled) ? s : 0 .................. :82-87
  * descriptor head convDa(3x3)+ReLU, convDb(1x1), F.normalize(p=2, dim=1) .......... :48-49, :88-89

Pinned against an import of that file in tests/golden/make_golden.py (bit-identical on
seeded weights; the import happens only in the build container, never on the GPU box).

The state dict is a plain ``dict[str, Tensor]`` with the reference's key layout
(conv1a.weight (64,1,3,3) ... convDb.bias (256,)).

``emulate_fp16=True`` rounds weights and every inter-layer activation to fp16 (fp32
accumulate) - the arithmetic contract of the HIP path (and of the reference's TensorRT
--fp16 engine, scripts/rebuild_engines.sh:88-97) - for tight-tolerance comparisons.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

ENC = ["conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b"]
POOL_AFTER = {"conv1b", "conv2b", "conv3b"}
SHAPES = {
    "conv1a": (64, 1, 3, 3), "conv1b": (64, 64, 3, 3),
    "conv2a": (64, 64, 3, 3), "conv2b": (64, 64, 3, 3),
    "conv3a": (128, 64, 3, 3), "conv3b": (128, 128, 3, 3),
    "conv4a": (128, 128, 3, 3), "conv4b": (128, 128, 3, 3),
    "convPa": (256, 128, 3, 3), "convPb": (65, 256, 1, 1),
    "convDa": (256, 128, 3, 3), "convDb": (256, 256, 1, 1),
}


def _q(t: torch.Tensor, on: bool) -> torch.Tensor:
    return t.half().float() if on else t


def _conv(sd, name, x, pad, q):
    return F.conv2d(x, _q(sd[name + ".weight"], q), sd[name + ".bias"], stride=1, padding=pad)


def encode(sd: dict, image: torch.Tensor, emulate_fp16: bool = False) -> torch.Tensor:
    """image [B,1,H,W] f32 in [0,1] -> features [B,128,H//8,W//8]."""
    q = emulate_fp16
    x = _q(image, q)
    for name in ENC:
        x = _q(F.relu(_conv(sd, name, x, 1, q)), q)
        if name in POOL_AFTER:
            x = F.max_pool2d(x, kernel_size=2, stride=2)
    return x


def detector_logits(sd, feat, emulate_fp16=False):
    """features -> raw 65-channel logits [B,65,Hc,Wc] (fp32, never rounded)."""
    q = emulate_fp16
    pa = _q(F.relu(_conv(sd, "convPa", feat, 1, q)), q)
    return _conv(sd, "convPb", pa, 0, q)


def heatmap_from_logits(logits: torch.Tensor) -> torch.Tensor:
    """softmax(65) -> drop dustbin -> depth-to-space x8: [B,65,Hc,Wc] -> [B,8Hc,8Wc]."""
    s = F.softmax(logits, 1)[:, :-1]
    b, _, h, w = s.shape
    s = s.permute(0, 2, 3, 1).reshape(b, h, w, 8, 8)
    return s.permute(0, 1, 3, 2, 4).reshape(b, h * 8, w * 8)


def nms(scores: torch.Tensor, radius: int = 4) -> torch.Tensor:
    """Single-pass max-pool NMS on [B,H,W]; plateaus all survive."""
    if radius <= 0:
        return scores
    s4 = scores.unsqueeze(1)
    pooled = F.max_pool2d(s4, 2 * radius + 1, stride=1, padding=radius)
    return torch.where(s4 == pooled, s4, torch.zeros_like(s4)).squeeze(1)


def descriptor_grid(sd, feat, emulate_fp16=False):
    """features -> L2-normalised dense descriptors [B,256,Hc,Wc] (fp32)."""
    q = emulate_fp16
    da = _q(F.relu(_conv(sd, "convDa", feat, 1, q)), q)
    d = _conv(sd, "convDb", da, 0, q)
    return F.normalize(d, p=2, dim=1)


def dense_forward(sd: dict, image: torch.Tensor, nms_radius: int = 4, emulate_fp16: bool = False):
    """Equivalent of DenseSuperPoint.forward: (scores [B,H8,W8] f32, descriptors [B,256,Hc,Wc])."""
    feat = encode(sd, image, emulate_fp16)
    logits = detector_logits(sd, feat, emulate_fp16)
    scores = nms(heatmap_from_logits(logits), nms_radius)
    desc = descriptor_grid(sd, feat, emulate_fp16)
    return scores, desc


def preprocess_u8(img_u8: torch.Tensor) -> torch.Tensor:
    """u8 [B,H,W] -> f32 [B,1,H,W] * (1/255)  (SuperPoint.cc:768-780: convertTo(CV_32F, 1.0/255.0)).

    OpenCV's 8u->32f convertTo works in float: dst = float(v) * float(alpha), one fp32 multiply.
    """
    return (img_u8.to(torch.float32) * torch.tensor(1.0 / 255.0, dtype=torch.float32)).unsqueeze(1)
