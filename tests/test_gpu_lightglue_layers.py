"""GPU (-m gpu): LightGlue internals through the test-only debug ABI (sship_lg_debug_*) against the fp64 oracle.

The end-to-end bars (matches0 agreement >= 0.99, |d mscores0| <= 2e-2) are only meaningful because every broken
variant of the algorithm violates them on these weights (tests/test_lightglue_known_answers.py, mutation table).  This
file adds the layer-by-layer view: normalised keypoints (bit-exact), rotary table, the residual stream after layers
1 / 5 / 9 and the assignment similarity, at N in {7x5, 64, 97x130, 600, 1024} and ragged lengths; plus full-size
properties (translation invariance, permutation equivariance, padding / batch-slot invariance).
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import hostpath as H  # noqa: E402
from oracle import lightglue_ref as LR  # noqa: E402

import sys as _sys  # noqa: E402
_sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import _lgcmp  # noqa: E402

X_REL_BAR = 4e-3      # ||x_gpu - x_ref|| / ||x_ref|| per sequence and layer (fp16 stream: ~1e-3 expected)
X_ABS_BAR = 6e-3      # max |x_gpu - x_ref| (x elements are O(0.1 .. 1))
SIM_REL_BAR = 4e-3    # max |sim_gpu - sim_ref| / max |sim_ref|
W, HH = 1376, 376


@pytest.fixture(scope="module")
def hip():
    from superslam_amd import _lib

    _lib.init()
    assert torch.cuda.is_available()
    return _lib.lib()


def _make_lg(weights_dir, max_kp, max_pairs=1):
    from superslam_amd import LightGlue

    m = LightGlue(weights_dir["lg_path"], W, HH, max_keypoints=max_kp, max_pairs=max_pairs)
    assert m.initialize(), m.last_error
    return m


@pytest.fixture(scope="module")
def lg(hip, weights_dir):
    m = _make_lg(weights_dir, 600, 2)
    yield m
    m.close()


def _px(k):
    """normalised [N,2] -> pixel coordinates whose normalisation returns k up to fp32 rounding."""
    s = max(W, HH) / 2.0
    return (np.asarray(k, np.float64) * s + np.array([W / 2.0, HH / 2.0])).astype(np.float32)


def _random_sets(n0, n1, seed):
    g = torch.Generator().manual_seed(seed)
    k0 = (torch.rand((n0, 2), generator=g) * 2 - 1) * torch.tensor([1.0, 0.27])
    d0 = torch.nn.functional.normalize(torch.randn((n0, 256), generator=g), dim=-1)
    perm = torch.randperm(max(n0, n1), generator=g)[:n1] % n0
    k1 = k0[perm] + 0.01 * torch.randn((n1, 2), generator=g)
    d1 = torch.nn.functional.normalize(d0[perm] + 0.15 * torch.randn((n1, 256), generator=g), dim=-1)
    return k0.numpy(), d0.half().float().numpy(), k1.numpy(), d1.half().float().numpy()


def _oracle(weights_dir, px0, d0, px1, d1, dtype=torch.float64):
    nk0, nk1 = H.normalize_kpts(px0, W, HH), H.normalize_kpts(px1, W, HH)
    with torch.no_grad():
        m, s, it = LR.match(weights_dir["lg"], torch.from_numpy(nk0)[None], torch.from_numpy(d0)[None],
                            torch.from_numpy(nk1)[None], torch.from_numpy(d1)[None], dtype=dtype, return_internals=True)
    return m[0].numpy(), s[0].numpy(), it


# ------------------------------------------------------------------------------------------------------
# a12: keypoint normalisation - host helper and device kernel, bit-exact against the committed table
# ------------------------------------------------------------------------------------------------------
def test_normalize_keypoints_host_and_device_bit_exact(hip, weights_dir, golden_dir):
    from superslam_amd import LightGlue

    with open(os.path.join(golden_dir, "meta.json")) as f:
        t = json.load(f)["normalize_kpts"]
    m = LightGlue(weights_dir["lg_path"], t["image_w"], t["image_h"], max_keypoints=64)
    assert m.initialize(), m.last_error
    kp = np.array(t["kp"], np.float32)
    exp = np.array(t["expected"], np.float32)
    np.testing.assert_array_equal(m.normalize_keypoints(kp), exp)                      # sship_lg_normalize_keypoints
    np.testing.assert_array_equal(m.normalize_keypoints(np.concatenate([kp, kp[:, :1]], 1)), exp)   # stride 3 (x, y, score)
    rng = np.random.default_rng(3)
    more = np.stack([rng.uniform(0, t["image_w"], 59), rng.uniform(0, t["image_h"], 59)], 1).astype(np.float32)
    allk = np.concatenate([kp, more])
    d = rng.standard_normal((len(allk), 256)).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    m.match(allk, d, allk[::-1].copy(), d[::-1].copy())
    ref = H.normalize_kpts(allk, t["image_w"], t["image_h"])
    got0 = m.debug_read(m.DEBUG_KPTS, 0, len(allk), 2)                                  # k_lg_prep on the device
    got1 = m.debug_read(m.DEBUG_KPTS, 1, len(allk), 2)
    np.testing.assert_array_equal(got0, ref)
    np.testing.assert_array_equal(got1, ref[::-1])
    np.testing.assert_array_equal(got0[: len(kp)], exp)
    # rotary table = (cos, sin)(Wr k) interleaved per frequency
    wr = weights_dir["lg"]["posenc.Wr.weight"].double().numpy()
    ph = ref.astype(np.float64) @ wr.T
    rope = m.debug_read(m.DEBUG_ROPE, 0, len(allk), 64)
    np.testing.assert_allclose(rope[:, 0::2], np.cos(ph), atol=3e-5)
    np.testing.assert_allclose(rope[:, 1::2], np.sin(ph), atol=3e-5)
    m.close()


# ------------------------------------------------------------------------------------------------------
# a14: residual stream and assignment similarity, layer by layer
# ------------------------------------------------------------------------------------------------------
def _check_layers(m, weights_dir, px0, d0, px1, d1, tag, parity_report, layers=(1, 5, 9), it=None, golden=None, x_abs_bar=X_ABS_BAR,
                  x_rel_bar=X_REL_BAR, sim_rel_bar=SIM_REL_BAR):
    n0, n1 = len(px0), len(px1)
    if it is None:
        _, _, it = _oracle(weights_dir, px0, d0, px1, d1)
    worst = 0.0
    for nl in layers:
        m.debug_set_layers(nl)
        m.match(px0, d0, px1, d1)
        for seq, (n, key) in enumerate(((n0, "x0_layers"), (n1, "x1_layers"))):
            got = m.debug_read(m.DEBUG_X, seq, n, 256)
            ref = it[key][nl - 1][0].double().numpy()
            rel = np.linalg.norm(got - ref) / np.linalg.norm(ref)
            mx = np.abs(got - ref).max()
            worst = max(worst, rel)
            print(f"LG layers {tag}: after layer {nl} seq {seq}: rel {rel:.2e} max|d| {mx:.2e} (|x| max {np.abs(ref).max():.2f})")
            assert np.isfinite(got).all(), (tag, nl, seq, "non-finite residual stream")
            assert rel <= x_rel_bar and mx <= x_abs_bar * max(1.0, np.abs(ref).max()), (tag, nl, seq, rel, mx)
            if golden is not None and nl in (1, 9):   # the committed fixture holds layers 0 and 8
                gref = golden[f"{tag}_x{seq}_l{nl - 1}"]
                assert np.linalg.norm(got - gref) / np.linalg.norm(gref) <= X_REL_BAR
    m.debug_set_layers(9)
    res = m.match(px0, d0, px1, d1)
    sim = m.debug_read(m.DEBUG_SIM, 0, n0, n1)
    sref = it["sim"][0].double().numpy()
    srel = np.abs(sim - sref).max() / np.abs(sref).max()
    print(f"LG layers {tag}: sim max|d| {np.abs(sim - sref).max():.3e} / max|sim| {np.abs(sref).max():.1f} = {srel:.2e}")
    assert np.isfinite(sim).all() and srel <= sim_rel_bar
    parity_report[f"lg_layers_{tag}"] = {"x_rel_worst": float(worst), "sim_rel": float(srel)}
    return res


@pytest.mark.parametrize("tag", ["n7x5", "n64x64", "n97x130"])
def test_layers_on_committed_fixtures(lg, weights_dir, golden_dir, parity_report, tag):
    g = np.load(os.path.join(golden_dir, "lightglue_selfcheck.npz"))
    px0, px1 = _px(g[tag + "_kpts0"]), _px(g[tag + "_kpts1"])
    d0, d1 = g[tag + "_desc0"].astype(np.float32), g[tag + "_desc1"].astype(np.float32)
    res = _check_layers(lg, weights_dir, px0, d0, px1, d1, tag, parity_report, golden=g)
    agree = (res.matches0 == g[tag + "_matches0"]).mean()
    ds = np.abs(res.mscores0 - g[tag + "_mscores0"]).max()
    print(f"LG {tag} vs committed vectors: agreement {agree:.4f} mscores max|d| {ds:.3e}")
    c = _lgcmp.compare(res.matches0, res.mscores0, g[tag + "_matches0"], g[tag + "_mscores0"])
    parity_report[f"lg_layers_{tag}"].update(c)
    _lgcmp.check(c)


@pytest.mark.parametrize("n0,n1,seed", [(600, 600, 31), (600, 17, 32), (33, 599, 33), (1, 1, 34)])
def test_layers_full_size_and_ragged(lg, weights_dir, parity_report, n0, n1, seed):
    k0, d0, k1, d1 = _random_sets(n0, n1, seed)
    px0, px1 = _px(k0), _px(k1)
    m_ref, s_ref, it = _oracle(weights_dir, px0, d0, px1, d1)
    tag = f"n{n0}x{n1}"
    res = _check_layers(lg, weights_dir, px0, d0, px1, d1, tag, parity_report, it=it)
    agree = (res.matches0 == m_ref).mean()
    ds = np.abs(res.mscores0 - s_ref).max()
    print(f"LG {tag}: matched ref {int((m_ref >= 0).sum())} got {int((res.matches0 >= 0).sum())} agreement {agree:.4f} max|d| {ds:.3e}")
    c = _lgcmp.compare(res.matches0, res.mscores0, m_ref, s_ref)
    parity_report[f"lg_layers_{tag}"].update(c)
    _lgcmp.check(c)


def test_layers_engine_max_1024(hip, weights_dir, parity_report):
    """The reference engine's upper profile (scripts/rebuild_engines.sh:118): 1024 keypoints per image."""
    m = _make_lg(weights_dir, 1024)
    k0, d0, k1, d1 = _random_sets(1024, 1000, 41)
    px0, px1 = _px(k0), _px(k1)
    m_ref, s_ref, it = _oracle(weights_dir, px0, d0, px1, d1)
    res = _check_layers(m, weights_dir, px0, d0, px1, d1, "n1024x1000", parity_report, layers=(1, 9), it=it)
    agree = (res.matches0 == m_ref).mean()
    ds = np.abs(res.mscores0 - s_ref).max()
    print(f"LG n1024x1000: agreement {agree:.4f} max|d| {ds:.3e}")
    c = _lgcmp.compare(res.matches0, res.mscores0, m_ref, s_ref)
    parity_report["lg_layers_n1024x1000"].update(c)
    _lgcmp.check(c)
    m.close()


# ------------------------------------------------------------------------------------------------------
# size-independent properties at full size
# ------------------------------------------------------------------------------------------------------
def test_translation_invariance(lg, parity_report):
    """Rotary self-attention sees relative positions only and cross-attention has no positional term: translating either
    keypoint set (independently) leaves matches0 unchanged and mscores0 within fp16 noise."""
    k0, d0, k1, d1 = _random_sets(600, 600, 51)
    px0, px1 = _px(k0 * 0.8), _px(k1 * 0.8)
    a = lg.match(px0, d0, px1, d1)
    b = lg.match(px0 + np.float32([37.0, -11.0]), d0, px1 + np.float32([-64.0, 23.0]), d1)
    agree = (a.matches0 == b.matches0).mean()
    ds = np.abs(a.mscores0 - b.mscores0).max()
    print(f"translation invariance: agreement {agree:.4f} max|d| {ds:.3e}")
    parity_report["lg_translation_invariance"] = {"agreement": float(agree), "mscores_maxd": float(ds)}
    assert agree >= 0.99 and ds <= _lgcmp.PATH_VS_PATH_BAR    # two fp16 runs with different rotary values: each within 2e-2 of the oracle
    assert (a.matches0 >= 0).sum() > 100


def test_permutation_equivariance(lg, parity_report):
    """Permuting set 1 permutes matches0's values; permuting set 0 permutes matches0 / mscores0 themselves."""
    k0, d0, k1, d1 = _random_sets(600, 577, 52)
    px0, px1 = _px(k0), _px(k1)
    a = lg.match(px0, d0, px1, d1)
    rng = np.random.default_rng(52)
    p1 = rng.permutation(577)
    b = lg.match(px0, d0, px1[p1], d1[p1])
    inv = np.empty_like(p1); inv[p1] = np.arange(577)
    exp = np.where(a.matches0 >= 0, inv[np.maximum(a.matches0, 0)], -1)
    agree1 = (b.matches0 == exp).mean()
    p0 = rng.permutation(600)
    c = lg.match(px0[p0], d0[p0], px1, d1)
    agree0 = (c.matches0 == a.matches0[p0]).mean()
    ds = max(np.abs(b.mscores0 - a.mscores0).max(), np.abs(c.mscores0 - a.mscores0[p0]).max())
    print(f"permutation equivariance: set-1 {agree1:.4f} set-0 {agree0:.4f} max|d| {ds:.3e}")
    parity_report["lg_permutation_equivariance"] = {"agreement": float(min(agree0, agree1)), "mscores_maxd": float(ds)}
    assert agree1 >= 0.99 and agree0 >= 0.99 and ds <= _lgcmp.PATH_VS_PATH_BAR


def test_padding_and_batch_slot_invariance(hip, lg, weights_dir):
    """The same problem gives the same answer in a handle with a larger max_keypoints (more padding tokens to mask) and in
    either slot of a two-pair batch next to an unrelated pair (ragged key masking, per-sequence lengths)."""
    k0, d0, k1, d1 = _random_sets(300, 211, 53)
    px0, px1 = _px(k0), _px(k1)
    a = lg.match(px0, d0, px1, d1)
    big = _make_lg(weights_dir, 1024)
    b = big.match(px0, d0, px1, d1)
    big.close()
    np.testing.assert_array_equal(a.matches0, b.matches0)
    np.testing.assert_allclose(a.mscores0, b.mscores0, atol=1e-6)
    # batch: pair 0 = this problem, pair 1 = an unrelated one of different sizes (and the other way round)
    o0, od0, o1, od1 = _random_sets(600, 555, 54)
    mk = 600

    def pack(sets):
        kp = torch.zeros((4, mk, 3), dtype=torch.float32)
        ds = torch.zeros((4, mk, 256), dtype=torch.float16)
        n = torch.zeros(4, dtype=torch.int32)
        for i, (k, d) in enumerate(sets):
            kp[i, : len(k), :2] = torch.from_numpy(k); ds[i, : len(k)] = torch.from_numpy(d).half(); n[i] = len(k)
        return kp.cuda(), n.cuda(), ds.cuda()

    for order in (0, 1):
        sets = [(px0, d0), (px1, d1), (_px(o0), od0), (_px(o1), od1)]
        if order:
            sets = sets[2:] + sets[:2]
        kp, n, ds = pack(sets)
        m0, ms0 = lg.match_batch_device(kp, n, ds)
        torch.cuda.synchronize()
        slot = 1 if order else 0
        np.testing.assert_array_equal(m0[slot, :300].cpu().numpy(), a.matches0)
        np.testing.assert_allclose(ms0[slot, :300].cpu().numpy(), a.mscores0, atol=1e-6)
        assert (m0[slot, 300:].cpu().numpy() == -1).all()


def test_counts_above_capacity_are_clamped(lg):
    """A device-side count above max_keypoints (a caller bug) is clamped, not followed past the sequence stride."""
    k0, d0, k1, d1 = _random_sets(600, 600, 55)
    kp = torch.zeros((2, 600, 3), dtype=torch.float32)
    kp[0, :, :2] = torch.from_numpy(_px(k0)); kp[1, :, :2] = torch.from_numpy(_px(k1))
    ds = torch.stack([torch.from_numpy(d0).half(), torch.from_numpy(d1).half()])
    ok = lg.match_batch_device(kp.cuda(), torch.tensor([600, 600], dtype=torch.int32).cuda(), ds.cuda())
    torch.cuda.synchronize()
    bad = lg.match_batch_device(kp.cuda(), torch.tensor([100000, 640], dtype=torch.int32).cuda(), ds.cuda())
    torch.cuda.synchronize()
    assert torch.equal(ok[0], bad[0]) and torch.allclose(ok[1], bad[1])


def test_loader_accepts_the_raw_checkpoint_key_layout(hip, lg, weights_dir, tmp_path):
    """self_attn.{i}.* / cross_attn.{i}.* (the published .pth layout) load to the same matcher as the module names."""
    from superslam_amd import LightGlue
    from superslam_amd.weights import save_safetensors, to_raw_checkpoint_keys

    raw_path = str(tmp_path / "lg_raw_names.safetensors")
    save_safetensors(to_raw_checkpoint_keys(weights_dir["lg"]), raw_path)
    m = LightGlue(raw_path, W, HH, max_keypoints=600, max_pairs=1)
    assert m.initialize(), m.last_error
    k0, d0, k1, d1 = _random_sets(200, 180, 61)
    a = lg.match(_px(k0), d0, _px(k1), d1)
    b = m.match(_px(k0), d0, _px(k1), d1)
    np.testing.assert_array_equal(a.matches0, b.matches0)
    np.testing.assert_array_equal(a.mscores0, b.mscores0)
    assert (a.matches0 >= 0).sum() > 20
    m.close()
    broken = {k: v for k, v in to_raw_checkpoint_keys(weights_dir["lg"]).items() if k != "cross_attn.3.to_v.weight"}
    save_safetensors(broken, raw_path)
    bad = LightGlue(raw_path, W, HH, max_keypoints=600)
    assert not bad.initialize() and "cross_attn.to_v.weight" in bad.last_error   # fails loudly, names the tensor


def test_throughput_batch_kernels_match_latency_kernels(hip, lg, weights_dir, parity_report):
    """A 64-pair batch runs the throughput kernels (4-wave FFN with two workgroups per CU, 2-tile attention); single calls
    run the latency kernels.  Same arithmetic, different tiling: every pair of the batch must agree with its single-call
    result (and through it with the oracle), including ragged lengths inside the batch."""
    P, mk = 64, 600
    big = _make_lg(weights_dir, mk, P)
    rng = np.random.default_rng(7)
    kp = torch.zeros((2 * P, mk, 3), dtype=torch.float32)
    ds = torch.zeros((2 * P, mk, 256), dtype=torch.float16)
    n = torch.zeros(2 * P, dtype=torch.int32)
    sets = []
    for p in range(P):
        n0, n1 = (600, 600) if p % 4 else (int(rng.integers(1, 601)), int(rng.integers(1, 601)))
        k0, d0, k1, d1 = _random_sets(n0, n1, 1000 + p)
        sets.append((_px(k0), d0, _px(k1), d1))
        for i, (k, d) in enumerate(((sets[-1][0], d0), (sets[-1][2], d1))):
            kp[2 * p + i, : len(k), :2] = torch.from_numpy(k); ds[2 * p + i, : len(k)] = torch.from_numpy(d).half(); n[2 * p + i] = len(k)
    m0, ms0 = big.match_batch_device(kp.cuda(), n.cuda(), ds.cuda())
    torch.cuda.synchronize()
    m0, ms0 = m0.cpu().numpy(), ms0.cpu().numpy()
    worst_agree, worst_ds, flips, rows = 1.0, 0.0, 0, 0
    for p in (0, 1, 4, 7, 8, 31, 32, 63):
        a = lg.match(*sets[p])
        n0 = len(sets[p][0])
        worst_agree = min(worst_agree, float((m0[p, :n0] == a.matches0).mean()))
        # mscores0 is exp(max) for mutual rows and 0 otherwise: a near-tie whose mutual flag flips between the two fp16 paths
        # moves the score by its full value - count those rows, compare the scores where the flag agrees
        same = (ms0[p, :n0] > 0) == (a.mscores0 > 0)
        flips += int((~same).sum()); rows += n0
        worst_ds = max(worst_ds, float(np.abs(ms0[p, :n0] - a.mscores0)[same].max()))
        assert (m0[p, n0:] == -1).all() and (ms0[p, n0:] == 0).all()
    assert flips <= 0.005 * rows, (flips, rows)
    # and one full-size pair of the batch straight against the oracle, layer by layer through the batch handle
    p = 5
    m_ref, s_ref, it = _oracle(weights_dir, *sets[p])
    x0 = big.debug_read(big.DEBUG_X, 2 * p, 600, 256)
    rel = np.linalg.norm(x0 - it["x0_layers"][8][0].double().numpy()) / np.linalg.norm(it["x0_layers"][8][0].double().numpy())
    agree = float((m0[p] == m_ref).mean())
    dso = float(np.abs(ms0[p] - s_ref).max())
    print(f"batch vs single: agreement {worst_agree:.4f} max|d| {worst_ds:.3e}; batch pair vs oracle: x rel {rel:.2e} agreement {agree:.4f} max|d| {dso:.3e}")
    parity_report["lg_batch64_vs_single"] = {"agreement_worst": worst_agree, "mscores_maxd": worst_ds, "x_rel_vs_oracle": float(rel),
                                             "agreement_vs_oracle": agree, "mscores_maxd_vs_oracle": dso}
    # two fp16 paths with different LayerNorm-statistics summation order: each is within 2e-2 of the oracle, so their mutual
    # distance is bounded by the sum
    assert worst_agree >= 0.99 and worst_ds <= _lgcmp.PATH_VS_PATH_BAR
    assert rel <= X_REL_BAR
    _lgcmp.check(_lgcmp.compare(m0[p], ms0[p], m_ref, s_ref))
    big.close()


# ------------------------------------------------------------------------------------------------------
# fp16 residual-stream headroom (VERDICT r02 "What's weak" 11): the published checkpoint cannot be loaded here, so the
# seeded weights are scaled until the residual stream is 70x .. 1000x larger than in the other tests (|x| up to ~45, ~170,
# ~670 after nine layers; fp16 tops out at 65 504) with the q / k gains scaled down by the same factor (attention logits
# stay O(40), as a trained matcher's do), and the same layer-by-layer comparison with the fp64 oracle must hold.
# The fourth case leaves the gains alone: scaled attention logits reach 8e4, beyond fp16's range (the reference's fp16 engine
# would already have overflowed in QK^T).  fp16 q / k (relative precision 5e-4) then carry an absolute logit error of ~40,
# the softmax is one-hot and flips between near-equal keys, so the fp64 oracle is no yardstick for single elements - what is
# asserted is that nothing becomes inf / NaN (the reference exponent of the attention kernel takes its two-slot form
# r = 256 a + b above |r| = 2048; a single fp16 r is off by up to 16 there and P = exp2(8 + 16) overflowed: found by this
# test - as was exp2(-r) = inf on a wave's first key tile when r < -128) and that the stream stays within 5 % of the oracle after
# the first layer.
# ------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("residual_gain,assign_gain,qk_scale,strict", [(4.0, 0.5, 1 / 40, True), (16.0, 0.12, 1 / 160, True),
                                                                      (64.0, 0.03, 1 / 700, True), (4.0, 0.5, 1.0, False)])
def test_large_residual_stream_magnitudes(hip, tmp_path, parity_report, residual_gain, assign_gain, qk_scale, strict):
    from superslam_amd import LightGlue
    from superslam_amd.weights import LG_CROSS_QK_GAIN, LG_SELF_QK_GAIN, make_lightglue_weights, save_safetensors

    sd = make_lightglue_weights(1, residual_gain=residual_gain, assign_gain=assign_gain,
                                self_qk_gain=tuple(v * qk_scale for v in LG_SELF_QK_GAIN),
                                cross_qk_gain=tuple(v * qk_scale for v in LG_CROSS_QK_GAIN))
    path = str(tmp_path / "lg_big.safetensors")
    save_safetensors(sd, path)
    m = LightGlue(path, W, HH, max_keypoints=160)
    assert m.initialize(), m.last_error
    k0, d0, k1, d1 = _random_sets(97, 130, 5)
    px0, px1 = _px(k0), _px(k1)
    m_ref, _, it = _oracle({"lg": sd}, px0, d0, px1, d1)
    xmax = float(it["x0_layers"][8].abs().max())
    tag = f"big_rg{int(residual_gain)}" + ("" if qk_scale < 1 else "_logits_beyond_fp16")
    # beyond the fp16 domain the error compounds chaotically with depth (rel 0.5 after five layers, for the classic softmax kernel
    # as well): the oracle is compared after the first layer only, everything after it must merely stay finite
    bars = {} if strict else {"x_abs_bar": 1.0, "x_rel_bar": 5e-2, "sim_rel_bar": float("inf"), "layers": (1,)}
    res = _check_layers(m, {"lg": sd}, px0, d0, px1, d1, tag, parity_report, it=it, **bars)
    assert np.isfinite(res.mscores0).all() and (res.mscores0 >= 0).all() and (res.mscores0 <= 1.0 + 1e-6).all()
    agree = float((res.matches0 == m_ref).mean())
    print(f"LG {tag}: |x| max after layer 9 = {xmax:.1f}, matches0 agreement {agree:.4f}")
    parity_report[f"lg_layers_{tag}"].update({"x_abs_max": xmax, "matches0_agreement": agree})
    assert xmax > 40.0 and (agree >= 0.97 or not strict)
    m.close()
