"""The LightGlue oracle and the EigenPlaces trunk against the independent implementations in `transformers` (CPU; oracle/pin_hf.py).

oracle/lightglue_ref.py restates cvg/LightGlue, which is neither under /root/reference nor in the image.  transformers
ships a port of the same model written by other people: with the same weights on the same inputs the two must agree.  The last
test breaks the oracle on purpose and requires the comparison to notice."""
import json
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pin_hf  # noqa: E402
from superslam_amd.weights import make_eigenplaces_weights, make_lightglue_weights  # noqa: E402

needs_lg = pytest.mark.skipif(not pin_hf.hf_lightglue_available(), reason="transformers without the LightGlue port")
needs_resnet = pytest.mark.skipif(not pin_hf.hf_resnet_available(), reason="transformers without ResNet")


@needs_lg
def test_lightglue_oracle_agrees_with_the_transformers_port(capsys):
    """Three committed fixtures (unequal keypoint counts go through the port's padding mask) + a seeded 300 x 280 problem with
    a second weight seed: matches0 identical, mscores0 and the residual stream after every one of the nine layers within 2e-6
    (the port's rotary / softmax run in fp32), the committed fixture equal to the port's output."""
    assert pin_hf.pin_lightglue(write=False) == 0, capsys.readouterr().out
    out = capsys.readouterr().out
    assert "PINNED against transformers' port" in out and out.count("0 matches0 differ") == 4, out


@needs_lg
def test_rekeying_sets_every_parameter_the_export_evaluates():
    sd = make_lightglue_weights(1)
    hf = pin_hf.lightglue_to_hf(sd)
    model, _ = pin_hf.build_hf_lightglue(sd)  # raises on unexpected / missing / mis-shaped keys
    own = model.state_dict()
    assert all(k in own for k in hf)
    # the fused upstream Wqkv really is split by (head, dim, {q, k, v}) and not by contiguous thirds
    w = sd["transformers.0.self_attn.Wqkv.weight"]
    assert torch.equal(hf["transformer_layers.0.self_attention.k_proj.weight"][65], w[(1 * 64 + 1) * 3 + 1])
    assert torch.equal(hf["transformer_layers.3.cross_attention.q_proj.weight"], hf["transformer_layers.3.cross_attention.k_proj.weight"])


@needs_lg
@pytest.mark.parametrize("mutation", ["rotate_half_sign", "rotary_not_interleaved", "qkv_contiguous", "self_scale_missing",
                                      "cross_scale_one_side", "cross_swapped_values", "single_log_softmax", "no_matchability",
                                      "gelu_tanh", "layernorm_no_affine"])
def test_a_broken_oracle_fails_the_pin(mutation):
    """Every listed mutation of oracle/lightglue_ref.py must move the comparison with the port beyond the pin's tolerance by a
    wide margin (>= 100x): the pin is able to fail on that step of the algorithm."""
    sd = make_lightglue_weights(1)
    model, _ = pin_hf.build_hf_lightglue(sd)
    gen = torch.Generator().manual_seed(3)
    k0 = torch.rand((96, 2), generator=gen, dtype=torch.float64) * 2 - 1
    k1 = torch.rand((96, 2), generator=gen, dtype=torch.float64) * 2 - 1  # equal counts: one of the mutations swaps the value tensors
    d0 = torch.nn.functional.normalize(torch.randn((96, 256), generator=gen, dtype=torch.float64), dim=-1)
    d1 = torch.nn.functional.normalize(torch.cat([d0[:60] + 0.05 * torch.randn((60, 256), generator=gen, dtype=torch.float64),
                                                  torch.randn((36, 256), generator=gen, dtype=torch.float64)]), dim=-1)
    ok = pin_hf.compare_lightglue(sd, model, torch, k0, d0, k1, d1)
    assert ok["matches_differ"] == 0 and ok["mscores_maxd"] < pin_hf.TOL and ok["layers_maxd"] < pin_hf.TOL and ok["matched"] >= 20, ok
    bad = pin_hf.compare_lightglue(sd, model, torch, k0, d0, k1, d1, mutations={mutation})
    assert bad["matches_differ"] > 0 or max(bad["mscores_maxd"], bad["layers_maxd"]) > 100 * pin_hf.TOL, (mutation, bad)


@needs_lg
def test_wrapper_normalisation_plus_patched_graph_equals_the_unpatched_port():
    """The reference patches the package's in-graph normalize_keypoints to a no-op (convert_lightglue_to_onnx.py:61) and normalises
    in its C++ wrapper instead (LightGlue.cc:241-251, restated in oracle/hostpath_ref.c).  The port fed PIXEL keypoints with its own,
    un-patched normalisation must give what the oracle gives on wrapper-normalised keypoints: same matches, mscores0 within 1e-5
    (the wrapper normalises in fp32)."""
    import numpy as np

    from oracle import hostpath
    from oracle import lightglue_ref as LR

    sd = make_lightglue_weights(1)
    model, _ = pin_hf.build_hf_lightglue(sd)
    gen = torch.Generator().manual_seed(21)
    n, w, h = 120, 1376, 376
    px0 = torch.rand((n, 2), generator=gen) * torch.tensor([w - 1.0, h - 1.0])
    px1 = (px0 + torch.tensor([-7.5, 0.25]) + 0.3 * torch.randn((n, 2), generator=gen)).clamp(min=0)
    d0 = torch.nn.functional.normalize(torch.randn((n, 256), generator=gen, dtype=torch.float64), dim=-1)
    d1 = torch.nn.functional.normalize(d0 + 0.05 * torch.randn((n, 256), generator=gen, dtype=torch.float64), dim=-1)
    m_hf, s_hf = pin_hf.run_hf_lightglue_pixels(model, torch, px0, d0, px1, d1, h, w)
    k0 = torch.from_numpy(hostpath.normalize_kpts(px0.numpy().astype(np.float32), w, h))
    k1 = torch.from_numpy(hostpath.normalize_kpts(px1.numpy().astype(np.float32), w, h))
    with torch.no_grad():
        m_ref, s_ref = LR.match(sd, k0[None], d0[None], k1[None], d1[None])
    assert int((m_hf >= 0).sum()) >= 60
    assert torch.equal(m_hf.to(torch.int32), m_ref[0]), int((m_hf.to(torch.int32) != m_ref[0]).sum())
    assert float((s_hf - s_ref[0].double()).abs().max()) < 1e-5


@needs_resnet
def test_eigenplaces_trunk_agrees_with_the_transformers_resnet18():
    """ResNet-18 trunk of oracle/eigenplaces_ref.py == transformers' ResNetModel (basic layers, 2-2-2-2) with the same weights,
    strict key match, two seeded inputs, fp64: identical feature maps."""
    worst, scale = pin_hf.compare_resnet18(make_eigenplaces_weights(2))
    assert worst <= 1e-10 and scale > 1.0, (worst, scale)
    # and it notices a structural error: stride-2 blocks evaluated with the stride on the second convolution instead of the first
    from oracle import eigenplaces_ref as ER
    orig = ER._block

    def wrong(sd, p, x, stride):
        import torch.nn.functional as F
        y = F.relu(ER._bn(sd, p + ".bn1", F.conv2d(x, sd[p + ".conv1.weight"], None, 1, 1)))
        y = ER._bn(sd, p + ".bn2", F.conv2d(y, sd[p + ".conv2.weight"], None, stride, 1))
        if (p + ".downsample.0.weight") in sd:
            x = ER._bn(sd, p + ".downsample.1", F.conv2d(x, sd[p + ".downsample.0.weight"], None, stride, 0))
        return F.relu(y + x)

    ER._block = wrong
    try:
        worst_bad, _ = pin_hf.compare_resnet18(make_eigenplaces_weights(2), seeds=(5,))
    finally:
        ER._block = orig
    assert worst_bad > 1e-3, worst_bad


def test_the_committed_stamp_names_what_was_pinned_against():
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "meta.json")))
    lg, ep = meta.get("lightglue_pinned_hf"), meta.get("eigenplaces_trunk_pinned_hf")
    assert lg and lg["matches0_identical"] and lg["mscores0_max_abs_dev"] < pin_hf.TOL and "transformers" in lg, lg
    assert ep and ep["feature_map_max_abs_dev_fp64"] <= 1e-10, ep
