"""CPU restatement of the LightGlue(features='superpoint') matcher (TEST INFRASTRUCTURE).

PARITY UNPINNED.  The arithmetic lives in the third-party package
``lightglue @ git+https://github.com/cvg/LightGlue.git`` (no tag/commit pinned:
/root/reference/utils/convert_lightglue_to_onnx.py:8), which is absent from /root/reference and
from this image, and none of the reference's tests hold a known-answer vector for
matches0/mscores0.  This file restates the published upstream ``lightglue/lightglue.py`` algorithm
 under the export-time overrides the reference applies:
  * in-graph normalize_keypoints patched to a no-op ......... convert_lightglue_to_onnx.py:61
  * flash = False, depth_confidence = width_confidence = -1 . :71-74  (all 9 layers, no pruning)
  * outputs matches0 -> int32 [1,N0], matching_scores0 [1,N0] :88-90
What IS pinned by reference-native code (keypoint normalisation, -1 filtering, dtypes) lives in
oracle/hostpath_ref.c and is golden-tested.

fp64 by default so it also serves as the high-precision anchor; ``dtype=torch.float32`` gives the
fp32 variant timed as cpu_baseline.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

N_LAYERS = 9
HEADS = 4
DIM = 256
HEAD_DIM = 64
FILTER_THRESHOLD = 0.1


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def posenc(sd, kpts):
    """LearnableFourierPositionalEncoding(2, 64, 64): Wr Linear(2->32, no bias) -> [2,B,1,N,64]."""
    proj = F.linear(kpts, sd["posenc.Wr.weight"])
    emb = torch.stack([torch.cos(proj), torch.sin(proj)], 0).unsqueeze(-3)
    return emb.repeat_interleave(2, dim=-1)


def rotate_half(x):
    x = x.unflatten(-1, (-1, 2))
    x1, x2 = x.unbind(dim=-1)
    return torch.stack((-x2, x1), dim=-1).flatten(start_dim=-2)


def apply_rotary(freqs, t):
    return t * freqs[0] + rotate_half(t) * freqs[1]


def _ffn(sd, p, x, msg):
    h = _lin(sd, p + "ffn.0", torch.cat([x, msg], -1))
    h = F.layer_norm(h, (2 * DIM,), sd[p + "ffn.1.weight"], sd[p + "ffn.1.bias"], 1e-5)
    h = F.gelu(h)
    return _lin(sd, p + "ffn.3", h)


def self_block(sd, i, x, enc):
    p = f"transformers.{i}.self_attn."
    qkv = _lin(sd, p + "Wqkv", x)
    qkv = qkv.unflatten(-1, (HEADS, -1, 3)).transpose(1, 2)  # [B,H,N,64,3]: q/k/v interleaved innermost
    q, k, v = qkv[..., 0], qkv[..., 1], qkv[..., 2]
    q = apply_rotary(enc, q)
    k = apply_rotary(enc, k)
    s = q.shape[-1] ** -0.5
    attn = F.softmax(torch.einsum("bhid,bhjd->bhij", q, k) * s, -1)
    ctx = torch.einsum("bhij,bhjd->bhid", attn, v)
    msg = _lin(sd, p + "out_proj", ctx.transpose(1, 2).flatten(start_dim=-2))
    return x + _ffn(sd, p, x, msg)


def cross_block(sd, i, x0, x1):
    p = f"transformers.{i}.cross_attn."

    def split(t):
        return t.unflatten(-1, (HEADS, -1)).transpose(1, 2)

    def merge(t):
        return t.transpose(1, 2).flatten(start_dim=-2)

    qk0, qk1 = split(_lin(sd, p + "to_qk", x0)), split(_lin(sd, p + "to_qk", x1))
    v0, v1 = split(_lin(sd, p + "to_v", x0)), split(_lin(sd, p + "to_v", x1))
    scale = HEAD_DIM ** -0.5
    qk0, qk1 = qk0 * scale ** 0.5, qk1 * scale ** 0.5
    sim = torch.einsum("bhid,bhjd->bhij", qk0, qk1)
    a01 = F.softmax(sim, dim=-1)
    a10 = F.softmax(sim.transpose(-2, -1).contiguous(), dim=-1)
    m0 = torch.einsum("bhij,bhjd->bhid", a01, v1)
    m1 = torch.einsum("bhji,bhjd->bhid", a10.transpose(-2, -1), v0)
    m0, m1 = _lin(sd, p + "to_out", merge(m0)), _lin(sd, p + "to_out", merge(m1))
    return x0 + _ffn(sd, p, x0, m0), x1 + _ffn(sd, p, x1, m1)


def log_assignment(sd, i, x0, x1):
    p = f"log_assignment.{i}."
    md0, md1 = _lin(sd, p + "final_proj", x0), _lin(sd, p + "final_proj", x1)
    d = md0.shape[-1]
    md0, md1 = md0 / d ** 0.25, md1 / d ** 0.25
    sim = torch.einsum("bmd,bnd->bmn", md0, md1)
    z0, z1 = _lin(sd, p + "matchability", x0), _lin(sd, p + "matchability", x1)
    cert = F.logsigmoid(z0) + F.logsigmoid(z1).transpose(1, 2)
    s0 = F.log_softmax(sim, 2)
    s1 = F.log_softmax(sim.transpose(-1, -2).contiguous(), 2).transpose(-1, -2)
    return s0 + s1 + cert, sim  # the [:m,:n] block; the dustbin row/col never reaches matches0/mscores0


def filter_matches(scores, th=FILTER_THRESHOLD):
    max0, max1 = scores.max(2), scores.max(1)
    m0, m1 = max0.indices, max1.indices
    idx0 = torch.arange(m0.shape[1])[None]
    mutual0 = idx0 == m1.gather(1, m0)
    max0_exp = max0.values.exp()
    mscores0 = torch.where(mutual0, max0_exp, max0_exp.new_tensor(0))
    valid0 = mutual0 & (mscores0 > th)
    m0 = torch.where(valid0, m0, m0.new_tensor(-1))
    return m0, mscores0


def match(sd: dict, kpts0, desc0, kpts1, desc1, dtype=torch.float64, return_internals: bool = False,
          n_layers: int = N_LAYERS):
    """kpts [1,N,2] ALREADY normalised, desc [1,N,256] -> (matches0 int32 [1,N0], mscores0 f32 [1,N0])."""
    sd = {k: v.to(dtype) for k, v in sd.items()}
    k0, k1, x0, x1 = (t.to(dtype) for t in (kpts0, kpts1, desc0, desc1))
    e0, e1 = posenc(sd, k0), posenc(sd, k1)
    for i in range(n_layers):
        x0 = self_block(sd, i, x0, e0)
        x1 = self_block(sd, i, x1, e1)
        x0, x1 = cross_block(sd, i, x0, x1)
    scores, sim = log_assignment(sd, N_LAYERS - 1, x0, x1)
    m0, ms0 = filter_matches(scores)
    if return_internals:
        return m0.to(torch.int32), ms0.to(torch.float32), dict(x0=x0, x1=x1, scores=scores, sim=sim)
    return m0.to(torch.int32), ms0.to(torch.float32)
