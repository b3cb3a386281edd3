"""Synthetic (seeded) weights + safetensors IO for the HIP front-end.

The reference's real weights are missing blobs (/root/reference/.MISSING_LARGE_BLOBS) and there
is no network, so parity and perf use shape-exact seeded state dicts with the reference's key
layout (SuperPoint: utils/convert_superpoint_to_onnx.py:38-49; LightGlue: upstream checkpoint
names).  Real weights dropped in as safetensors load through the same path
(the reference ships utils/export_safetensors.py for the .pth -> safetensors step).
"""
from __future__ import annotations

import math

import torch

SP_SHAPES = {
    "conv1a": (64, 1, 3, 3), "conv1b": (64, 64, 3, 3),
    "conv2a": (64, 64, 3, 3), "conv2b": (64, 64, 3, 3),
    "conv3a": (128, 64, 3, 3), "conv3b": (128, 128, 3, 3),
    "conv4a": (128, 128, 3, 3), "conv4b": (128, 128, 3, 3),
    "convPa": (256, 128, 3, 3), "convPb": (65, 256, 1, 1),
    "convDa": (256, 128, 3, 3), "convDb": (256, 256, 1, 1),
}

LG_LAYERS = 9
LG_DIM = 256
LG_HEADS = 4


def make_superpoint_weights(seed: int = 0, peak_gain: float = 8.0) -> dict:
    """He-uniform conv weights from a CPU generator (bit-reproducible across machines).

    ``peak_gain`` scales convPb so the 65-way softmax is peaked enough for the 9x9 NMS and
    the 0.005 threshold to be meaningful (a default-init head gives a near-uniform heatmap
    whose every pixel is > 0.005 - SURVEY.md 'Hard parts').
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}
    for name, shp in SP_SHAPES.items():
        fan_in = shp[1] * shp[2] * shp[3]
        bound = math.sqrt(6.0 / fan_in)  # gain sqrt(2) he-uniform keeps ReLU activations O(1)
        w = (torch.rand(shp, generator=g, dtype=torch.float32) * 2 - 1) * bound
        b = (torch.rand(shp[0], generator=g, dtype=torch.float32) * 2 - 1) * 0.05
        if name == "convPb":
            w = w * peak_gain
        sd[name + ".weight"] = w.contiguous()
        sd[name + ".bias"] = b.contiguous()
    return sd


def _lin(g, out_f, in_f, gain=1.0, bias_scale=0.02):
    bound = gain * math.sqrt(3.0 / in_f)
    w = (torch.rand((out_f, in_f), generator=g, dtype=torch.float32) * 2 - 1) * bound
    b = (torch.rand((out_f,), generator=g, dtype=torch.float32) * 2 - 1) * bias_scale
    return w.contiguous(), b.contiguous()


# Per-layer gains on the q/k rows of SelfBlock.Wqkv and on CrossBlock.to_qk.  With he-style unit-variance rows and a
# unit-norm descriptor stream the attention logits would have std ~0.004, i.e. every softmax would be uniform and a
# parity test could not tell attention from averaging (round-1 VERDICT, "What's weak" 1).  These constants were
# calibrated once (scripts/calibrate_lg_gains.py) so that the logits of every layer have std ~2 on the n97x130 fixture;
# they are literals, not computed at import, so the state dict stays bit-reproducible across machines.
LG_SELF_QK_GAIN = (22.0, 18.0, 16.0, 14.0, 13.0, 12.0, 11.0, 10.0, 9.8)
LG_CROSS_QK_GAIN = (18.0, 15.0, 14.0, 12.0, 11.0, 10.0, 9.5, 9.5, 8.7)
LG_POSENC_GAIN = 6.0


def make_lightglue_weights(seed: int = 1, residual_gain: float = 0.05, assign_gain: float = 24.0,
                           self_qk_gain=LG_SELF_QK_GAIN, cross_qk_gain=LG_CROSS_QK_GAIN,
                           posenc_gain: float = LG_POSENC_GAIN) -> dict:
    """Seeded LightGlue(features='superpoint') state dict, upstream key layout.

    ``residual_gain`` keeps each block's update small against the unit-norm descriptor stream and
    ``assign_gain`` makes final_proj a scaled near-identity, so the random-weight matcher behaves
    like a sharpened mutual-nearest-neighbour matcher: synthetic stereo pairs then produce a
    realistic number of confident matches instead of an all -1 output.  ``self_qk_gain`` /
    ``cross_qk_gain`` / ``posenc_gain`` make the attention softmaxes peaked and position dependent
    (see the constants above) so that errors in QK^T, the softmax, rotary or the keypoint
    normalisation move matches0 / mscores0 by more than the parity tolerances.
    """
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}
    sd["posenc.Wr.weight"] = (torch.randn((32, 2), generator=g, dtype=torch.float32) * posenc_gain).contiguous()
    d = LG_DIM
    for i in range(LG_LAYERS):
        p = f"transformers.{i}.self_attn."
        w, b = _lin(g, 3 * d, d)
        # rows are (head, dim, {q,k,v}) - unflatten(-1, (4, 64, 3)): scale the q and k rows only
        w.view(LG_HEADS, d // LG_HEADS, 3, d)[:, :, 0:2] *= self_qk_gain[i]
        b.view(LG_HEADS, d // LG_HEADS, 3)[:, :, 0:2] *= self_qk_gain[i]
        sd[p + "Wqkv.weight"], sd[p + "Wqkv.bias"] = w, b
        sd[p + "out_proj.weight"], sd[p + "out_proj.bias"] = _lin(g, d, d)
        sd[p + "ffn.0.weight"], sd[p + "ffn.0.bias"] = _lin(g, 2 * d, 2 * d)
        sd[p + "ffn.1.weight"] = (1.0 + 0.1 * torch.randn(2 * d, generator=g)).contiguous()
        sd[p + "ffn.1.bias"] = (0.05 * torch.randn(2 * d, generator=g)).contiguous()
        sd[p + "ffn.3.weight"], sd[p + "ffn.3.bias"] = _lin(g, d, 2 * d, gain=residual_gain, bias_scale=0.002)
        p = f"transformers.{i}.cross_attn."
        sd[p + "to_qk.weight"], sd[p + "to_qk.bias"] = _lin(g, d, d, gain=cross_qk_gain[i], bias_scale=0.02 * cross_qk_gain[i])
        sd[p + "to_v.weight"], sd[p + "to_v.bias"] = _lin(g, d, d)
        sd[p + "to_out.weight"], sd[p + "to_out.bias"] = _lin(g, d, d)
        sd[p + "ffn.0.weight"], sd[p + "ffn.0.bias"] = _lin(g, 2 * d, 2 * d)
        sd[p + "ffn.1.weight"] = (1.0 + 0.1 * torch.randn(2 * d, generator=g)).contiguous()
        sd[p + "ffn.1.bias"] = (0.05 * torch.randn(2 * d, generator=g)).contiguous()
        sd[p + "ffn.3.weight"], sd[p + "ffn.3.bias"] = _lin(g, d, 2 * d, gain=residual_gain, bias_scale=0.002)
    # Only log_assignment[LG_LAYERS-1] is used with depth_confidence = -1
    # (utils/convert_lightglue_to_onnx.py:71-74); the others exist in the checkpoint.
    for i in range(LG_LAYERS):
        p = f"log_assignment.{i}."
        w, b = _lin(g, d, d, gain=0.2)
        sd[p + "final_proj.weight"] = (w + assign_gain * torch.eye(d)).contiguous()
        sd[p + "final_proj.bias"] = b
        sd[p + "matchability.weight"], sd[p + "matchability.bias"] = _lin(g, 1, d, gain=1.0)
        sd[p + "matchability.bias"] = sd[p + "matchability.bias"] + 2.0
    return sd


def normalize_lightglue_keys(sd: dict) -> dict:
    """Raw upstream checkpoint names -> module names: ``self_attn.{i}.*`` -> ``transformers.{i}.self_attn.*`` (and
    cross_attn), optional ``matcher.`` prefix dropped - the rename upstream's LightGlue.__init__ applies at load time
    (checkpoint key layout).  The C loader (sship_lg_weights_load) applies the same rule."""
    import re

    out = {}
    for k, v in sd.items():
        if k.startswith("matcher."):
            k = k[len("matcher."):]
        k = re.sub(r"^(self_attn|cross_attn)\.(\d+)\.", lambda m: f"transformers.{m.group(2)}.{m.group(1)}.", k)
        out[k] = v
    return out


def to_raw_checkpoint_keys(sd: dict) -> dict:
    """Inverse of normalize_lightglue_keys (tests: a state dict in the published checkpoint's layout)."""
    import re

    return {re.sub(r"^transformers\.(\d+)\.(self_attn|cross_attn)\.", lambda m: f"{m.group(2)}.{m.group(1)}.", k): v
            for k, v in sd.items()}


def make_eigenplaces_weights(seed: int = 2) -> dict:
    """Seeded EigenPlaces(ResNet18, 512) state dict with the hub model's key layout (backbone = nn.Sequential of torchvision's
    ResNet-18 children without avgpool / fc, aggregation = [L2Norm, GeM, Flatten, Linear, L2Norm]; oracle/eigenplaces_ref.py).
    Conv weights are he-normal, BatchNorm statistics are near identity with a little spread so that folding them matters."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    sd = {}

    def conv(name, cout, cin, k):
        sd[name + ".weight"] = (torch.randn((cout, cin, k, k), generator=g) * math.sqrt(2.0 / (cin * k * k))).contiguous()

    def bn(name, c, gamma=1.0):
        sd[name + ".weight"] = (gamma * (1.0 + 0.1 * torch.randn(c, generator=g))).contiguous()
        sd[name + ".bias"] = (0.05 * torch.randn(c, generator=g)).contiguous()
        sd[name + ".running_mean"] = (0.05 * torch.randn(c, generator=g)).contiguous()
        sd[name + ".running_var"] = (1.0 + 0.2 * torch.rand(c, generator=g)).contiguous()
        sd[name + ".num_batches_tracked"] = torch.tensor(1, dtype=torch.int64)

    conv("backbone.0", 64, 3, 7); bn("backbone.1", 64)
    cin = 64
    for idx, planes, stride in ((4, 64, 1), (5, 128, 2), (6, 256, 2), (7, 512, 2)):
        for b in range(2):
            p = f"backbone.{idx}.{b}"
            conv(p + ".conv1", planes, cin if b == 0 else planes, 3); bn(p + ".bn1", planes)
            conv(p + ".conv2", planes, planes, 3); bn(p + ".bn2", planes, gamma=0.5)   # damped residual branch: activations stay O(1)
            if b == 0 and (stride != 1 or cin != planes):
                conv(p + ".downsample.0", planes, cin, 1); bn(p + ".downsample.1", planes)
        cin = planes
    sd["aggregation.1.p"] = torch.tensor([3.0])
    w, b = _lin(g, 512, 512, gain=1.0, bias_scale=0.02)
    sd["aggregation.3.weight"], sd["aggregation.3.bias"] = w, b
    return sd


def save_safetensors(sd: dict, path: str) -> None:
    from safetensors.torch import save_file

    save_file({k: v.contiguous() for k, v in sd.items()}, path)


def load_safetensors(path: str) -> dict:
    from safetensors.torch import load_file

    return load_file(path)


def state_dict_sha256(sd: dict) -> str:
    """SHA-256 over the raw little-endian bytes of every tensor in sorted-key order."""
    import hashlib

    h = hashlib.sha256()
    for k in sorted(sd):
        h.update(k.encode())
        h.update(sd[k].contiguous().numpy().tobytes())
    return h.hexdigest()
