"""CPU: compensating evidence for the parity-unpinned LightGlue oracle (oracle/lightglue_ref.py).

The upstream ``lightglue`` package is absent (utils/convert_lightglue_to_onnx.py:8), so nothing from the reference
pins matches0 / mscores0.  What this file pins instead:

* per-function known answers written out BY HAND from the published algorithm: the positional
  encoding layout, rotate_half's pairing and sign, the ``unflatten(-1, (4, 64, 3))`` q/k/v interleave, the both-sides
  cross-attention scale, the double log-softmax assignment, filter_matches - none of the expected values below are
  produced by calling the oracle;
* a complete hand-computed N0 = N1 = 3 matcher pass through an independent numpy transcription (identity-like
  weights so every intermediate is a short closed form);
* the MUTATION TABLE: every deliberately broken variant of the oracle must move matches0 / mscores0 on the committed
  fixtures by more than the GPU parity tolerances (agreement >= 0.99, |d mscores| <= 2e-2) - i.e. the GPU parity
  suite is able to fail on that step (round-1 VERDICT "What's weak" 1: with the old weights uniform attention passed).
"""
import math
import os

import numpy as np
import pytest
import torch

from oracle import lightglue_ref as L
from superslam_amd.weights import LG_DIM, LG_LAYERS, make_lightglue_weights

GPU_AGREEMENT_BAR = 0.99   # tests/test_gpu_parity.py: matches0 row agreement
GPU_MSCORE_BAR = 2e-2      # tests/test_gpu_parity.py: max |mscores0 - oracle|
GPU_X_REL_BAR = 4e-3       # tests/test_gpu_lightglue_layers.py: ||x_gpu - x_oracle|| / ||x_oracle|| per layer


# ------------------------------------------------------------------------------------------------------
# per-function known answers
# ------------------------------------------------------------------------------------------------------
def test_posenc_layout_known_answer():
    """LearnableFourierPositionalEncoding(2, 64, 64): p = Wr k (32 values); emb[0] = cos p, emb[1] = sin p, each value
    repeated twice along the last axis (repeat_interleave): [c0, c0, c1, c1, ...], shape [2, B, 1, N, 64]."""
    wr = torch.zeros((32, 2), dtype=torch.float64)
    wr[0] = torch.tensor([1.0, 0.0])     # p0 = x
    wr[1] = torch.tensor([0.0, 2.0])     # p1 = 2 y
    wr[31] = torch.tensor([1.0, 1.0])    # p31 = x + y
    k = torch.tensor([[[math.pi / 2, math.pi / 4]]], dtype=torch.float64)  # one keypoint
    e = L.posenc({"posenc.Wr.weight": wr}, k)
    assert tuple(e.shape) == (2, 1, 1, 1, 64)
    cos, sin = e[0, 0, 0, 0].numpy(), e[1, 0, 0, 0].numpy()
    exp_p = np.zeros(32)
    exp_p[0], exp_p[1], exp_p[31] = math.pi / 2, math.pi / 2, 3 * math.pi / 4
    np.testing.assert_allclose(cos, np.repeat(np.cos(exp_p), 2), atol=1e-12)
    np.testing.assert_allclose(sin, np.repeat(np.sin(exp_p), 2), atol=1e-12)
    assert cos[0] == cos[1] and cos[62] == cos[63] and abs(cos[62] + math.sqrt(0.5)) < 1e-12   # interleaved, not tiled


def test_rotate_half_and_rotary_known_answer():
    """rotate_half pairs (x[2i], x[2i+1]) -> (-x[2i+1], x[2i]); apply = t*cos + rotate_half(t)*sin, i.e. every
    consecutive pair is rotated by +theta_i: a 90 degree rotation maps (1, 0) -> (0, 1)."""
    t = torch.tensor([1.0, 2.0, 3.0, 4.0], dtype=torch.float64)
    assert L.rotate_half(t).tolist() == [-2.0, 1.0, -4.0, 3.0]
    theta = torch.tensor([math.pi / 2, math.pi / 2, math.pi, math.pi], dtype=torch.float64)   # already interleaved
    out = L.apply_rotary(torch.stack([torch.cos(theta), torch.sin(theta)]), torch.tensor([1.0, 0.0, 1.0, 2.0], dtype=torch.float64))
    np.testing.assert_allclose(out.numpy(), [0.0, 1.0, -1.0, -2.0], atol=1e-12)
    # relative-position property the rotary form exists for: <R(a) q, R(b) k> depends on a - b only
    q, k = torch.tensor([0.3, -1.2], dtype=torch.float64), torch.tensor([0.7, 0.4], dtype=torch.float64)

    def rot(v, a):
        a = torch.tensor([a, a], dtype=torch.float64)
        return L.apply_rotary(torch.stack([torch.cos(a), torch.sin(a)]), v)

    assert abs(float(rot(q, 0.9) @ rot(k, 0.5)) - float(rot(q, 0.4) @ k)) < 1e-12


def test_wqkv_interleave_known_answer():
    """Wqkv output feature f = head*192 + dim*3 + c with c in {q, k, v}: unflatten(-1, (4, 64, 3))."""
    f = torch.arange(768, dtype=torch.float64)[None, None]          # [B=1, N=1, 768], value = feature index
    q, k, v = L.split_qkv(f)
    assert tuple(q.shape) == (1, 4, 1, 64)
    for h in (0, 3):
        for d in (0, 1, 63):
            assert q[0, h, 0, d].item() == h * 192 + d * 3 + 0
            assert k[0, h, 0, d].item() == h * 192 + d * 3 + 1
            assert v[0, h, 0, d].item() == h * 192 + d * 3 + 2


def _zero_block_weights(p, d=256):
    z = torch.zeros
    return {p + "ffn.0.weight": z(2 * d, 2 * d), p + "ffn.0.bias": z(2 * d), p + "ffn.1.weight": torch.ones(2 * d),
            p + "ffn.1.bias": z(2 * d), p + "ffn.3.weight": z(d, 2 * d), p + "ffn.3.bias": z(d)}


def test_self_block_known_answer_attention_average():
    """One SelfBlock with zero q/k weights (uniform attention), v = x[:, :64] copied into head 0 and an FFN that adds
    msg[0] into x[1]: a closed form that exercises Wqkv row order, softmax, out_proj and the cat[x, msg] order."""
    d, n = 256, 3
    p = "transformers.0.self_attn."
    sd = _zero_block_weights(p)
    wqkv = torch.zeros(768, d)
    for dd in range(64):
        wqkv[0 * 192 + dd * 3 + 2, dd] = 1.0            # v of head 0, dim dd <- x[dd]
    sd[p + "Wqkv.weight"], sd[p + "Wqkv.bias"] = wqkv, torch.zeros(768)
    sd[p + "out_proj.weight"], sd[p + "out_proj.bias"] = torch.eye(d), torch.zeros(d)
    # ffn: h = LN(W0 cat[x, msg]) ... make the FFN a plain probe of msg: not linear because of LN/GELU, so instead
    # check the message path directly with the oracle's building blocks
    sd = {k: v.double() for k, v in sd.items()}
    x = torch.zeros(1, n, d, dtype=torch.float64)
    x[0, 0, 0], x[0, 1, 0], x[0, 2, 0] = 3.0, 6.0, 9.0
    q, k, v = L.split_qkv(L._lin(sd, p + "Wqkv", x))
    assert q.abs().sum() == 0 and k.abs().sum() == 0
    assert v[0, 0, :, 0].tolist() == [3.0, 6.0, 9.0] and v[0, 1:].abs().sum() == 0
    attn = torch.softmax(torch.einsum("bhid,bhjd->bhij", q, k) / 8, -1)
    np.testing.assert_allclose(attn.numpy(), 1.0 / 3.0)
    ctx = torch.einsum("bhij,bhjd->bhid", attn, v).transpose(1, 2).flatten(start_dim=-2)
    np.testing.assert_allclose(ctx[0, :, 0].numpy(), [6.0, 6.0, 6.0])   # mean of (3, 6, 9) lands in channel 0 (head 0, dim 0)
    assert ctx[0, :, 1:].abs().sum() == 0


def test_cross_block_scale_both_sides_known_answer():
    """Non-flash CrossBlock: qk0 and qk1 are EACH scaled by 64^-0.25, so sim = <qk0, qk1> / 8."""
    d = 256
    p = "transformers.0.cross_attn."
    sd = _zero_block_weights(p)
    sd[p + "to_qk.weight"], sd[p + "to_qk.bias"] = torch.eye(d), torch.zeros(d)
    sd[p + "to_v.weight"], sd[p + "to_v.bias"] = torch.eye(d), torch.zeros(d)
    sd[p + "to_out.weight"], sd[p + "to_out.bias"] = torch.eye(d), torch.zeros(d)
    sd = {k: v.double() for k, v in sd.items()}
    x0 = torch.zeros(1, 1, d, dtype=torch.float64); x1 = torch.zeros(1, 2, d, dtype=torch.float64)
    x0[0, 0, 0] = 4.0
    x1[0, 0, 0], x1[0, 1, 0] = 2.0, 0.0           # head-0 logits: 4*2/8 = 1 and 0
    x1[0, 0, 1], x1[0, 1, 1] = 10.0, 20.0         # value channel 1 (head 0)
    # ffn.3 = 0 -> the block returns x unchanged; probe the message with the oracle's pieces
    y0, y1 = L.cross_block(sd, 0, x0, x1)
    assert torch.equal(y0, x0) and torch.equal(y1, x1)
    w = math.e / (math.e + 1.0)                   # softmax([1, 0])[0]
    qk0 = x0.unflatten(-1, (4, -1)).transpose(1, 2) * 64 ** -0.25
    qk1 = x1.unflatten(-1, (4, -1)).transpose(1, 2) * 64 ** -0.25
    sim = torch.einsum("bhid,bhjd->bhij", qk0, qk1)
    np.testing.assert_allclose(sim[0, 0, 0].numpy(), [1.0, 0.0], atol=1e-12)
    m0 = torch.softmax(sim, -1)[0, 0, 0] @ x1[0, :, 1]
    assert abs(m0.item() - (10.0 * w + 20.0 * (1 - w))) < 1e-12


def test_log_assignment_known_answer():
    """md = final_proj(x) / 256^0.25 ; sim = md0 md1^T ; S = log_softmax_rows + log_softmax_cols + logsig(z0) + logsig(z1)^T."""
    d = 256
    p = "log_assignment.8."
    sd = {p + "final_proj.weight": 4.0 * torch.eye(d), p + "final_proj.bias": torch.zeros(d),
          p + "matchability.weight": torch.zeros(1, d), p + "matchability.bias": torch.tensor([0.0])}
    sd = {k: v.double() for k, v in sd.items()}
    x0 = torch.zeros(1, 2, d, dtype=torch.float64); x1 = torch.zeros(1, 2, d, dtype=torch.float64)
    x0[0, 0, 0] = x0[0, 1, 1] = 1.0
    x1[0, 0, 0] = x1[0, 1, 1] = 2.0
    scores, sim = L.log_assignment(sd, 8, x0, x1)
    # md0 = x0 * 4 / 4 = x0, md1 = x1  ->  sim = [[2, 0], [0, 2]]
    np.testing.assert_allclose(sim[0].numpy(), [[2.0, 0.0], [0.0, 2.0]], atol=1e-12)
    lsm_diag = 2.0 - math.log(math.exp(2.0) + 1.0)
    lsm_off = 0.0 - math.log(math.exp(2.0) + 1.0)
    cert = 2.0 * math.log(0.5)                    # logsigmoid(0) twice
    np.testing.assert_allclose(scores[0].numpy(), [[2 * lsm_diag + cert, 2 * lsm_off + cert],
                                                   [2 * lsm_off + cert, 2 * lsm_diag + cert]], atol=1e-12)
    m0, ms0 = L.filter_matches(scores)
    assert m0[0].tolist() == [0, 1]
    np.testing.assert_allclose(ms0[0].numpy(), [math.exp(2 * lsm_diag + cert)] * 2, rtol=1e-6)   # 0.1939 > 0.1


def test_filter_matches_known_answer():
    """mutual arg-max, mscores0 = exp(max) for mutual rows (also below the threshold), matches0 = -1 unless mutual and > 0.1."""
    s = torch.log(torch.tensor([[[0.90, 0.05, 0.01],
                                 [0.05, 0.08, 0.02],      # mutual with column 1 but 0.08 <= 0.1 -> -1, score kept
                                 [0.85, 0.01, 0.30]]], dtype=torch.float64))   # best column 0 belongs to row 0 -> not mutual
    m0, ms0 = L.filter_matches(s)
    assert m0[0].tolist() == [0, -1, -1]
    np.testing.assert_allclose(ms0[0].numpy(), [0.90, 0.08, 0.0], rtol=1e-6)


# ------------------------------------------------------------------------------------------------------
# a complete matcher pass by an independent transcription (numpy, loops over heads; no code shared with the oracle)
# ------------------------------------------------------------------------------------------------------
def _np_lightglue(sd, k0, d0, k1, d1, n_layers):
    sd = {k: v.double().numpy() for k, v in sd.items()}

    def lin(n, x):
        return x @ sd[n + ".weight"].T + sd[n + ".bias"]

    def enc(k):
        p = k @ sd["posenc.Wr.weight"].T                      # [N, 32]
        return np.repeat(np.cos(p), 2, axis=1), np.repeat(np.sin(p), 2, axis=1)   # [N, 64]

    def rope(t, cs):                                         # t [N, 64]
        c, s = cs
        r = np.empty_like(t)
        r[:, 0::2] = -t[:, 1::2]
        r[:, 1::2] = t[:, 0::2]
        return t * c + r * s

    def softmax(a):
        a = a - a.max(-1, keepdims=True)
        e = np.exp(a)
        return e / e.sum(-1, keepdims=True)

    def ffn(p, x, msg):
        h = lin(p + "ffn.0", np.concatenate([x, msg], -1))
        mu, var = h.mean(-1, keepdims=True), h.var(-1, keepdims=True)
        h = (h - mu) / np.sqrt(var + 1e-5) * sd[p + "ffn.1.weight"] + sd[p + "ffn.1.bias"]
        h = 0.5 * h * (1.0 + np.vectorize(math.erf)(h / math.sqrt(2.0)))
        return lin(p + "ffn.3", h)

    def self_block(i, x, cs):
        p = f"transformers.{i}.self_attn."
        qkv = lin(p + "Wqkv", x)                              # [N, 768]
        ctx = np.zeros_like(x)
        for h in range(4):
            q = np.stack([qkv[:, h * 192 + dd * 3 + 0] for dd in range(64)], 1)
            k = np.stack([qkv[:, h * 192 + dd * 3 + 1] for dd in range(64)], 1)
            v = np.stack([qkv[:, h * 192 + dd * 3 + 2] for dd in range(64)], 1)
            a = softmax(rope(q, cs) @ rope(k, cs).T / 8.0)
            ctx[:, h * 64:(h + 1) * 64] = a @ v
        return x + ffn(p, x, lin(p + "out_proj", ctx))

    def cross_block(i, x0, x1):
        p = f"transformers.{i}.cross_attn."
        qk0, qk1, v0, v1 = lin(p + "to_qk", x0), lin(p + "to_qk", x1), lin(p + "to_v", x0), lin(p + "to_v", x1)
        m0, m1 = np.zeros_like(x0), np.zeros_like(x1)
        for h in range(4):
            sl = slice(h * 64, (h + 1) * 64)
            sim = (qk0[:, sl] * 64 ** -0.25) @ (qk1[:, sl] * 64 ** -0.25).T
            m0[:, sl] = softmax(sim) @ v1[:, sl]
            m1[:, sl] = softmax(sim.T) @ v0[:, sl]
        return x0 + ffn(p, x0, lin(p + "to_out", m0)), x1 + ffn(p, x1, lin(p + "to_out", m1))

    x0, x1, e0, e1 = d0.copy(), d1.copy(), enc(k0), enc(k1)
    for i in range(n_layers):
        x0, x1 = self_block(i, x0, e0), self_block(i, x1, e1)
        x0, x1 = cross_block(i, x0, x1)
    p = "log_assignment.8."
    sim = (lin(p + "final_proj", x0) / 4.0) @ (lin(p + "final_proj", x1) / 4.0).T
    ls = lambda z: -np.log1p(np.exp(-z))                      # noqa: E731  logsigmoid
    z0, z1 = lin(p + "matchability", x0), lin(p + "matchability", x1)
    lsm = lambda a: a - np.log(np.exp(a - a.max(-1, keepdims=True)).sum(-1, keepdims=True)) - a.max(-1, keepdims=True)  # noqa: E731
    S = lsm(sim) + lsm(sim.T).T + ls(z0) + ls(z1).T
    j = S.argmax(1)
    mutual = S.argmax(0)[j] == np.arange(len(j))
    ms = np.where(mutual, np.exp(S.max(1)), 0.0)
    return np.where(mutual & (ms > 0.1), j, -1), ms, x0, x1, sim


@pytest.mark.parametrize("n0,n1,seed", [(3, 3, 5), (9, 6, 6)])
def test_oracle_equals_independent_transcription(n0, n1, seed):
    """The full 9-layer matcher, oracle (torch, batched einsum) vs the loop-per-head numpy transcription above."""
    sd = make_lightglue_weights(1)
    g = torch.Generator().manual_seed(seed)
    k0 = (torch.rand((n0, 2), generator=g, dtype=torch.float64) * 2 - 1) * torch.tensor([1.0, 0.27], dtype=torch.float64)
    k1 = (torch.rand((n1, 2), generator=g, dtype=torch.float64) * 2 - 1) * torch.tensor([1.0, 0.27], dtype=torch.float64)
    d0 = torch.nn.functional.normalize(torch.randn((n0, 256), generator=g, dtype=torch.float64), dim=-1)
    d1 = torch.nn.functional.normalize(torch.randn((n1, 256), generator=g, dtype=torch.float64), dim=-1)
    d1[: min(n0, n1)] = torch.nn.functional.normalize(d0[: min(n0, n1)] + 0.1 * d1[: min(n0, n1)], dim=-1)
    with torch.no_grad():
        m, s, it = L.match(sd, k0[None], d0[None], k1[None], d1[None], return_internals=True)
    m_np, s_np, x0_np, x1_np, sim_np = _np_lightglue(sd, k0.numpy(), d0.numpy(), k1.numpy(), d1.numpy(), LG_LAYERS)
    np.testing.assert_allclose(it["x0"][0].numpy(), x0_np, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(it["x1"][0].numpy(), x1_np, rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(it["sim"][0].numpy(), sim_np, rtol=1e-9, atol=1e-10)
    np.testing.assert_array_equal(m[0].numpy(), m_np)
    np.testing.assert_allclose(s[0].numpy(), s_np, atol=1e-6)
    assert (m_np >= 0).sum() >= 1


def test_hand_computed_three_by_three():
    """N0 = N1 = 3, weights chosen so that every layer is the identity on x (ffn.3 = 0) and final_proj = 8 I,
    matchability z = 3: matches and scores follow from the descriptors' dot products by hand.

      d0 = e0, e1, e2 ;  d1 = e1, e0, (e2 + e3)/sqrt 2
      md = 8 x / 4 = 2 x  ->  sim = 4 <d0_i, d1_j> = [[0,4,0],[4,0,0],[0,0,2.828]]
    """
    d = LG_DIM
    sd = {"posenc.Wr.weight": torch.zeros(32, 2)}
    for i in range(LG_LAYERS):
        for blk, names in (("self_attn", ("Wqkv", "out_proj")), ("cross_attn", ("to_qk", "to_v", "to_out"))):
            p = f"transformers.{i}.{blk}."
            sd.update(_zero_block_weights(p))
            for nme in names:
                rows = 768 if nme == "Wqkv" else d
                sd[p + nme + ".weight"], sd[p + nme + ".bias"] = 0.01 * torch.ones(rows, d), torch.zeros(rows)
        sd[f"log_assignment.{i}.final_proj.weight"], sd[f"log_assignment.{i}.final_proj.bias"] = 8.0 * torch.eye(d), torch.zeros(d)
        sd[f"log_assignment.{i}.matchability.weight"] = torch.zeros(1, d)
        sd[f"log_assignment.{i}.matchability.bias"] = torch.tensor([3.0])
    e = torch.eye(d, dtype=torch.float64)
    d0 = torch.stack([e[0], e[1], e[2]])[None]
    d1 = torch.stack([e[1], e[0], (e[2] + e[3]) / math.sqrt(2.0)])[None]
    k = torch.zeros(1, 3, 2, dtype=torch.float64)
    with torch.no_grad():
        m, s, it = L.match(sd, k, d0, k, d1, return_internals=True)
    r2 = 4.0 / math.sqrt(2.0)
    np.testing.assert_allclose(it["sim"][0].numpy(), [[0, 4, 0], [4, 0, 0], [0, 0, r2]], atol=1e-12)
    assert m[0].tolist() == [1, 0, 2]
    lsig = -math.log1p(math.exp(-3.0))
    row01 = 4.0 - math.log(math.exp(4.0) + 2.0)                 # log-softmax of the 4 in rows 0/1 (and columns 0/1)
    row2 = r2 - math.log(math.exp(r2) + 2.0)
    expect = [math.exp(2 * row01 + 2 * lsig), math.exp(2 * row01 + 2 * lsig), math.exp(2 * row2 + 2 * lsig)]
    np.testing.assert_allclose(s[0].numpy(), expect, rtol=1e-6)   # 0.8444, 0.8444, 0.7257 (computed by hand)
    # 4 - ln(e^4 + 2) = -0.03597, logsigmoid(3) = -0.04859: exp(-0.16911) = 0.84440 ; 2.8284 - ln(e^2.8284 + 2) = -0.11173: exp(-0.32063) = 0.72569
    np.testing.assert_allclose(expect, [0.84440, 0.84440, 0.72569], atol=2e-5)


# ------------------------------------------------------------------------------------------------------
# the mutation table
# ------------------------------------------------------------------------------------------------------
# mutation -> which bar it must break: "match" (matches0 agreement / mscores0) or "x" (per-layer residual stream)
MUTATION_BARS = {
    "uniform_self_attention": "match", "uniform_cross_attention": "match", "no_rotary": "match",
    "rotate_half_sign": "match", "rotary_not_interleaved": "match", "qkv_contiguous": "match",
    "self_scale_missing": "match", "cross_scale_one_side": "match", "cross_swapped_values": "match",
    "single_log_softmax": "match", "no_matchability": "match", "layernorm_no_affine": "match",
    # tanh-GELU differs from erf-GELU by < 5e-4 per element: below what an fp16 residual stream can resolve.  It is
    # listed to document that limit (asserted below): the GPU's GELU (A&S 7.1.26 erf, |err| <= 1.5e-7) cannot be told
    # from the exact one by any end-to-end test either.
    "gelu_tanh": "none",
}


def _fixture(golden_dir, tag):
    g = np.load(os.path.join(golden_dir, "lightglue_selfcheck.npz"))
    t = lambda a: torch.from_numpy(np.asarray(a, np.float32))[None]   # noqa: E731
    return g, (t(g[tag + "_kpts0"]), t(g[tag + "_desc0"]), t(g[tag + "_kpts1"]), t(g[tag + "_desc1"]))


def test_mutation_table_is_complete():
    assert set(MUTATION_BARS) == set(L.MUTATIONS)


@pytest.mark.parametrize("mutation", [m for m in L.MUTATIONS])
def test_every_mutation_breaks_the_gpu_parity_bars(golden_dir, mutation):
    sd = make_lightglue_weights(1)
    worst_agree, worst_ds, worst_x = 1.0, 0.0, 0.0
    for tag in ("n64x64", "n97x130"):
        if mutation == "cross_swapped_values" and tag != "n64x64":
            continue
        g, f = _fixture(golden_dir, tag)
        with torch.no_grad():
            m, s, it = L.match(sd, *f, mutations={mutation}, return_internals=True)
        worst_agree = min(worst_agree, float((m[0].numpy() == g[tag + "_matches0"]).mean()))
        worst_ds = max(worst_ds, float(np.abs(s[0].numpy() - g[tag + "_mscores0"]).max()))
        x8 = it["x0_layers"][8][0].float().numpy()
        worst_x = max(worst_x, float(np.linalg.norm(x8 - g[tag + "_x0_l8"]) / np.linalg.norm(g[tag + "_x0_l8"])))
    print(f"{mutation}: agreement {worst_agree:.3f}  max|d mscores| {worst_ds:.3f}  x rel {worst_x:.2e}")
    bar = MUTATION_BARS[mutation]
    if bar == "match":
        assert worst_agree < GPU_AGREEMENT_BAR or worst_ds > GPU_MSCORE_BAR
        assert worst_ds > 5 * GPU_MSCORE_BAR       # not marginal: an order of magnitude above the tolerance
    else:
        assert worst_agree == 1.0 and worst_ds < GPU_MSCORE_BAR and worst_x < GPU_X_REL_BAR   # documented blind spot


def test_keypoint_normalisation_errors_break_the_bars(golden_dir):
    """Input-side mutations of LightGlue.cc:241-251: per-axis scaling, min instead of max, x/y swapped."""
    sd = make_lightglue_weights(1)
    tag = "n97x130"
    g, (k0, d0, k1, d1) = _fixture(golden_dir, tag)
    W, Hh = 1376.0, 376.0
    s = max(W, Hh) / 2

    def px(k):
        return k * s + torch.tensor([W / 2, Hh / 2])

    variants = {
        "per_axis_scale": lambda p: (p - torch.tensor([W / 2, Hh / 2])) / torch.tensor([W / 2, Hh / 2]),
        "xy_swapped": lambda p: ((p - torch.tensor([W / 2, Hh / 2])) / s).flip(-1),
        "min_instead_of_max": lambda p: (p - torch.tensor([W / 2, Hh / 2])) / (min(W, Hh) / 2),
    }
    for name, fn in variants.items():
        with torch.no_grad():
            m, sc = L.match(sd, fn(px(k0)).float(), d0, fn(px(k1)).float(), d1)
        agree = float((m[0].numpy() == g[tag + "_matches0"]).mean())
        ds = float(np.abs(sc[0].numpy() - g[tag + "_mscores0"]).max())
        print(f"kpt normalisation '{name}': agreement {agree:.3f} max|d| {ds:.3f}")
        assert agree < GPU_AGREEMENT_BAR or ds > GPU_MSCORE_BAR, name
    # Missing centring is NOT detectable, by construction: rotary self-attention sees relative positions only and the
    # cross-attention has no positional term, so the matcher is invariant to a translation of either keypoint set.
    # (The GPU suite uses that as a full-size property test; the centring itself is pinned bit-exactly by the
    # normalize_kpts table in meta.json.)
    with torch.no_grad():
        m, sc = L.match(sd, (px(k0) / s).float(), d0, (px(k1) / s + 0.3).float(), d1)
    assert float((m[0].numpy() == g[tag + "_matches0"]).mean()) == 1.0
    assert float(np.abs(sc[0].numpy() - g[tag + "_mscores0"]).max()) < 1e-4
    # sanity: the correct normalisation round-trips to the fixture exactly enough
    with torch.no_grad():
        m, sc = L.match(sd, ((px(k0) - torch.tensor([W / 2, Hh / 2])) / s).float(), d0,
                        ((px(k1) - torch.tensor([W / 2, Hh / 2])) / s).float(), d1)
    assert float((m[0].numpy() == g[tag + "_matches0"]).mean()) == 1.0


def test_unmasked_padding_breaks_the_bars(golden_dir):
    """Ragged key masking: appending zero-descriptor padding tokens WITHOUT masking them (what a kernel that ignores
    `lens` would do) changes the result beyond the bars; the oracle on the unpadded sets is the truth."""
    sd = make_lightglue_weights(1)
    tag = "n97x130"
    g, (k0, d0, k1, d1) = _fixture(golden_dir, tag)
    pad = lambda t, n: torch.cat([t, torch.zeros(1, n - t.shape[1], t.shape[2])], 1)   # noqa: E731
    with torch.no_grad():
        m, s = L.match(sd, pad(k0, 128), pad(d0, 128), pad(k1, 160), pad(d1, 160))
    m, s = m[0, :97].numpy(), s[0, :97].numpy()
    agree = float((m == g[tag + "_matches0"]).mean())
    ds = float(np.abs(s - g[tag + "_mscores0"]).max())
    print(f"unmasked padding: agreement {agree:.3f} max|d| {ds:.3f}")
    assert agree < GPU_AGREEMENT_BAR or ds > GPU_MSCORE_BAR


def test_raw_checkpoint_key_layout_round_trip():
    """The published checkpoint names blocks self_attn.{i}.* / cross_attn.{i}.*; upstream renames them at load."""
    from superslam_amd.weights import normalize_lightglue_keys, to_raw_checkpoint_keys

    sd = make_lightglue_weights(1)
    raw = to_raw_checkpoint_keys(sd)
    assert "self_attn.0.Wqkv.weight" in raw and "cross_attn.8.to_qk.bias" in raw and "posenc.Wr.weight" in raw
    assert not any(k.startswith("transformers.") for k in raw)
    back = normalize_lightglue_keys({("matcher." + k if i % 2 else k): v for i, (k, v) in enumerate(raw.items())})
    assert set(back) == set(sd)
    assert all(torch.equal(back[k], sd[k]) for k in sd)
    assert normalize_lightglue_keys({"self_attn.12.ffn.0.weight": 1}) == {"transformers.12.self_attn.ffn.0.weight": 1}


def test_kernel_gelu_constants_against_erf():
    """The HIP kernels evaluate GELU as y * sigmoid(y Q(y^2)) (csrc/lg_ffn.h, gelu2): parse the constants from the kernel
    source, evaluate the same formula in fp32 and compare with the exact erf form over the whole useful range."""
    import re

    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "superslam_amd", "csrc", "lg_ffn.h")).read()
    q = [np.float32(re.search(rf"kGeluQ{i} = (-?[0-9.]+e[+-][0-9]+)f", src).group(1)) for i in range(5)]
    y = np.linspace(-12, 12, 200001).astype(np.float32)
    s = y * y
    p = ((((s * q[4] + q[3]) * s + q[2]) * s + q[1]) * s + q[0])
    with np.errstate(over="ignore"):
        out = y * (np.float32(1) / (np.float32(1) + np.exp2(y * p)))
    exact = torch.nn.functional.gelu(torch.from_numpy(y).double()).numpy()
    err = np.abs(out.astype(np.float64) - exact)
    print(f"kernel GELU vs erf: max abs {err.max():.2e}, max rel (|gelu| >= 0.05) {(err / np.maximum(np.abs(exact), 5e-2)).max():.2e}")
    assert err.max() <= 1.0e-5
    assert (err / np.maximum(np.abs(exact), 5e-2)).max() <= 6e-5       # < 1/4 of half an fp16 ulp (2.4e-4)
    assert np.isfinite(out).all() and out[0] <= 0 and abs(out[-1] - 12.0) < 1e-6
    # far tails: saturates to y / -0, never NaN
    big = np.array([-1e4, -100.0, 100.0, 1e4], np.float32)
    sb = big * big
    pb = ((((sb * q[4] + q[3]) * sb + q[2]) * sb + q[1]) * sb + q[0])
    with np.errstate(over="ignore", invalid="ignore"):
        ob = big * (np.float32(1) / (np.float32(1) + np.exp2(big * pb)))
    assert np.isfinite(ob).all() and ob[0] == 0 and ob[1] == 0 and ob[2] == 100.0 and ob[3] == 1e4
