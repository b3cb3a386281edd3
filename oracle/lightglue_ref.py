"""CPU restatement of the LightGlue(features='superpoint') matcher (TEST INFRASTRUCTURE).

PARITY: PINNED AGAINST AN INDEPENDENT PORT, UNPINNED AGAINST THE PACKAGE THE REFERENCE IMPORTS.
The arithmetic lives in the third-party package
``lightglue @ git+https://github.com/cvg/LightGlue.git`` (no tag/commit pinned:
/root/reference/utils/convert_lightglue_to_onnx.py:8), which is absent from /root/reference and
from this image, and none of the reference's tests hold a known-answer vector for
matches0/mscores0.  What the image does hold is transformers' port of the same model
(transformers.models.lightglue, 5.15.0): oracle/pin_hf.py re-keys the seeded weights into it and
tests/test_oracle_pins_hf.py asserts identical matches0, mscores0 and every layer's residual stream
within 2e-6 (the port's fp32 rotary / softmax), and that ten mutations of this file break the
comparison.  oracle/pin_oracles.py is the remaining check against the cvg package itself.  This file restates the published upstream ``lightglue/lightglue.py`` algorithm
 under the export-time overrides the reference applies:
  * in-graph normalize_keypoints patched to a no-op ......... convert_lightglue_to_onnx.py:61
  * flash = False, depth_confidence = width_confidence = -1 . :71-74  (all 9 layers, no pruning)
  * outputs matches0 -> int32 [1,N0], matching_scores0 [1,N0] :88-90
What IS pinned by reference-native code (keypoint normalisation, -1 filtering, dtypes) lives in
oracle/hostpath_ref.c and is golden-tested.  Compensating evidence for the unpinned part:
per-function known answers and a hand-computed 3x3 case in tests/test_lightglue_known_answers.py.

fp64 by default so it also serves as the high-precision anchor; ``dtype=torch.float32`` gives the
fp32 variant timed as cpu_baseline.

``mutations`` deliberately BREAKS one step of the algorithm (tests/test_oracle_golden.py's mutation
table: every mutation must move matches0 / mscores0 by more than the GPU parity tolerances, i.e. the
parity suite is able to fail on that step).  The un-mutated path is the oracle.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

N_LAYERS = 9
HEADS = 4
DIM = 256
HEAD_DIM = 64
FILTER_THRESHOLD = 0.1

MUTATIONS = (
    "uniform_self_attention",    # softmax(QK^T) replaced by 1/N averaging in every SelfBlock
    "uniform_cross_attention",   # same in every CrossBlock
    "no_rotary",                 # q, k used without the positional rotation
    "rotate_half_sign",          # rotate_half returns (x2, -x1) instead of (-x2, x1)
    "rotary_not_interleaved",    # cos/sin tiled [c0..c31, c0..c31] instead of repeat_interleave(2)
    "qkv_contiguous",            # Wqkv output split as (3, 4, 64) instead of unflatten(-1, (4, 64, 3))
    "self_scale_missing",        # q k^T without the 1/sqrt(64)
    "cross_scale_one_side",      # only one side of qk scaled by 64^-0.25
    "cross_swapped_values",      # m0 <- v0 / m1 <- v1 (messages built from the wrong image)
    "single_log_softmax",        # log_assignment with the row log-softmax only
    "no_matchability",           # logsigmoid(z) terms dropped
    "gelu_tanh",                 # tanh-approximate GELU instead of the exact erf form
    "layernorm_no_affine",       # LayerNorm gamma / beta ignored
)


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def posenc(sd, kpts, mutations=frozenset()):
    """LearnableFourierPositionalEncoding(2, 64, 64): Wr Linear(2->32, no bias) -> [2,B,1,N,64]."""
    proj = F.linear(kpts, sd["posenc.Wr.weight"])
    emb = torch.stack([torch.cos(proj), torch.sin(proj)], 0).unsqueeze(-3)
    if "rotary_not_interleaved" in mutations:
        return torch.cat([emb, emb], dim=-1)
    return emb.repeat_interleave(2, dim=-1)


def rotate_half(x, mutations=frozenset()):
    x = x.unflatten(-1, (-1, 2))
    x1, x2 = x.unbind(dim=-1)
    if "rotate_half_sign" in mutations:
        return torch.stack((x2, -x1), dim=-1).flatten(start_dim=-2)
    return torch.stack((-x2, x1), dim=-1).flatten(start_dim=-2)


def apply_rotary(freqs, t, mutations=frozenset()):
    if "no_rotary" in mutations:
        return t
    return t * freqs[0] + rotate_half(t, mutations) * freqs[1]


def _ffn(sd, p, x, msg, mutations=frozenset()):
    h = _lin(sd, p + "ffn.0", torch.cat([x, msg], -1))
    if "layernorm_no_affine" in mutations:
        h = F.layer_norm(h, (2 * DIM,), None, None, 1e-5)
    else:
        h = F.layer_norm(h, (2 * DIM,), sd[p + "ffn.1.weight"], sd[p + "ffn.1.bias"], 1e-5)
    h = F.gelu(h, approximate="tanh") if "gelu_tanh" in mutations else F.gelu(h)
    return _lin(sd, p + "ffn.3", h)


def split_qkv(qkv, mutations=frozenset()):
    """Wqkv output [B,N,768] -> q, k, v [B,H,N,64]: q/k/v are interleaved innermost (unflatten(-1, (4, 64, 3)))."""
    if "qkv_contiguous" in mutations:
        t = qkv.unflatten(-1, (3, HEADS, -1)).permute(2, 0, 3, 1, 4)  # [3,B,H,N,64]
        return t[0], t[1], t[2]
    qkv = qkv.unflatten(-1, (HEADS, -1, 3)).transpose(1, 2)  # [B,H,N,64,3]
    return qkv[..., 0], qkv[..., 1], qkv[..., 2]


def self_block(sd, i, x, enc, mutations=frozenset()):
    p = f"transformers.{i}.self_attn."
    q, k, v = split_qkv(_lin(sd, p + "Wqkv", x), mutations)
    q = apply_rotary(enc, q, mutations)
    k = apply_rotary(enc, k, mutations)
    s = 1.0 if "self_scale_missing" in mutations else q.shape[-1] ** -0.5
    attn = F.softmax(torch.einsum("bhid,bhjd->bhij", q, k) * s, -1)
    if "uniform_self_attention" in mutations:
        attn = torch.full_like(attn, 1.0 / attn.shape[-1])
    ctx = torch.einsum("bhij,bhjd->bhid", attn, v)
    msg = _lin(sd, p + "out_proj", ctx.transpose(1, 2).flatten(start_dim=-2))
    return x + _ffn(sd, p, x, msg, mutations)


def cross_block(sd, i, x0, x1, mutations=frozenset()):
    p = f"transformers.{i}.cross_attn."

    def split(t):
        return t.unflatten(-1, (HEADS, -1)).transpose(1, 2)

    def merge(t):
        return t.transpose(1, 2).flatten(start_dim=-2)

    qk0, qk1 = split(_lin(sd, p + "to_qk", x0)), split(_lin(sd, p + "to_qk", x1))
    v0, v1 = split(_lin(sd, p + "to_v", x0)), split(_lin(sd, p + "to_v", x1))
    scale = HEAD_DIM ** -0.5
    qk0 = qk0 * scale ** 0.5
    if "cross_scale_one_side" not in mutations:
        qk1 = qk1 * scale ** 0.5
    sim = torch.einsum("bhid,bhjd->bhij", qk0, qk1)
    a01 = F.softmax(sim, dim=-1)
    a10 = F.softmax(sim.transpose(-2, -1).contiguous(), dim=-1)
    if "uniform_cross_attention" in mutations:
        a01 = torch.full_like(a01, 1.0 / a01.shape[-1])
        a10 = torch.full_like(a10, 1.0 / a10.shape[-1])
    if "cross_swapped_values" in mutations:
        v0, v1 = v1[:, :, : v0.shape[2]] if v1.shape[2] >= v0.shape[2] else v1, v0  # only used with n0 == n1
    m0 = torch.einsum("bhij,bhjd->bhid", a01, v1)
    m1 = torch.einsum("bhji,bhjd->bhid", a10.transpose(-2, -1), v0)
    m0, m1 = _lin(sd, p + "to_out", merge(m0)), _lin(sd, p + "to_out", merge(m1))
    return x0 + _ffn(sd, p, x0, m0, mutations), x1 + _ffn(sd, p, x1, m1, mutations)


def log_assignment(sd, i, x0, x1, mutations=frozenset()):
    p = f"log_assignment.{i}."
    md0, md1 = _lin(sd, p + "final_proj", x0), _lin(sd, p + "final_proj", x1)
    d = md0.shape[-1]
    md0, md1 = md0 / d ** 0.25, md1 / d ** 0.25
    sim = torch.einsum("bmd,bnd->bmn", md0, md1)
    z0, z1 = _lin(sd, p + "matchability", x0), _lin(sd, p + "matchability", x1)
    cert = F.logsigmoid(z0) + F.logsigmoid(z1).transpose(1, 2)
    if "no_matchability" in mutations:
        cert = torch.zeros_like(cert)
    s0 = F.log_softmax(sim, 2)
    s1 = F.log_softmax(sim.transpose(-1, -2).contiguous(), 2).transpose(-1, -2)
    if "single_log_softmax" in mutations:
        s1 = torch.zeros_like(s1)
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
          n_layers: int = N_LAYERS, mutations=frozenset()):
    """kpts [1,N,2] ALREADY normalised, desc [1,N,256] -> (matches0 int32 [1,N0], mscores0 f32 [1,N0]).

    return_internals adds a dict with the residual streams after every layer (``x0_layers`` / ``x1_layers``,
    lists of [1,N,256]), the final ``x0`` / ``x1``, the assignment ``sim`` [1,N0,N1] and ``scores``."""
    mutations = frozenset(mutations)
    assert mutations <= frozenset(MUTATIONS), mutations - frozenset(MUTATIONS)
    sd = {k: v.to(dtype) for k, v in sd.items()}
    k0, k1, x0, x1 = (t.to(dtype) for t in (kpts0, kpts1, desc0, desc1))
    e0, e1 = posenc(sd, k0, mutations), posenc(sd, k1, mutations)
    xs0, xs1 = [], []
    for i in range(n_layers):
        x0 = self_block(sd, i, x0, e0, mutations)
        x1 = self_block(sd, i, x1, e1, mutations)
        x0, x1 = cross_block(sd, i, x0, x1, mutations)
        xs0.append(x0)
        xs1.append(x1)
    scores, sim = log_assignment(sd, N_LAYERS - 1, x0, x1, mutations)
    m0, ms0 = filter_matches(scores)
    if return_internals:
        return m0.to(torch.int32), ms0.to(torch.float32), dict(x0=x0, x1=x1, scores=scores, sim=sim, x0_layers=xs0,
                                                               x1_layers=xs1, enc0=e0, enc1=e1)
    return m0.to(torch.int32), ms0.to(torch.float32)
