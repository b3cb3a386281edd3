#!/usr/bin/env python3
"""One-off calibration of weights.LG_{SELF,CROSS}_QK_GAIN (build container, CPU): walks the fp64 oracle layer by layer on
the n97x130 fixture inputs and picks, per layer, the q/k gain that gives attention logits of std ~= TARGET, rounded to two
significant digits.  The result is pasted into superslam_amd/weights.py as literals (bit-reproducible weights)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import lightglue_ref as L  # noqa: E402
from superslam_amd import weights as W  # noqa: E402

TARGET = 2.0


def fixture(n0, n1, seed):
    g = torch.Generator().manual_seed(seed)
    k0 = (torch.rand((1, n0, 2), generator=g) * 2 - 1) * torch.tensor([1.0, 0.27])
    d0 = F.normalize(torch.randn((1, n0, 256), generator=g), dim=-1)
    perm = torch.randperm(max(n0, n1), generator=g)[:n1] % n0
    k1 = k0[:, perm] + 0.01 * torch.randn((1, n1, 2), generator=g)
    d1 = F.normalize(d0[:, perm] + 0.15 * torch.randn((1, n1, 256), generator=g), dim=-1)
    return k0.double(), d0.half().double(), k1.double(), d1.half().double()


def layer_stats(sd, i, x0, x1, e0, e1):
    p = f"transformers.{i}.self_attn."
    qkv = L._lin(sd, p + "Wqkv", x0).unflatten(-1, (4, -1, 3)).transpose(1, 2)
    q, k = L.apply_rotary(e0, qkv[..., 0]), L.apply_rotary(e0, qkv[..., 1])
    lg = torch.einsum("bhid,bhjd->bhij", q, k) / 8
    y0, y1 = L.self_block(sd, i, x0, e0), L.self_block(sd, i, x1, e1)
    pc = f"transformers.{i}.cross_attn."
    a = L._lin(sd, pc + "to_qk", y0).unflatten(-1, (4, -1)).transpose(1, 2) * 64 ** -0.25
    b = L._lin(sd, pc + "to_qk", y1).unflatten(-1, (4, -1)).transpose(1, 2) * 64 ** -0.25
    sim = torch.einsum("bhid,bhjd->bhij", a, b)
    return lg, sim, y0, y1


def r2(v):
    return float(f"{v:.2g}")


def main():
    k0, d0, k1, d1 = fixture(97, 130, 23)
    sg, cg = [1.0] * 9, [1.0] * 9
    for i in range(9):
        for _ in range(2):  # self gain, then cross gain (the cross input depends on the self gain)
            sd = {k: v.double() for k, v in W.make_lightglue_weights(1, self_qk_gain=sg, cross_qk_gain=cg).items()}
            e0, e1 = L.posenc(sd, k0), L.posenc(sd, k1)
            x0, x1 = d0, d1
            for j in range(i):
                x0, x1 = L.self_block(sd, j, x0, e0), L.self_block(sd, j, x1, e1)
                x0, x1 = L.cross_block(sd, j, x0, x1)
            lg, sim, _, _ = layer_stats(sd, i, x0, x1, e0, e1)
            sg[i] = r2(sg[i] * (TARGET / lg.std().item()) ** 0.5)
            cg[i] = r2(cg[i] * (TARGET / sim.std().item()) ** 0.5)
        print(f"layer {i}: self gain {sg[i]} cross gain {cg[i]}")
    print("LG_SELF_QK_GAIN =", tuple(sg))
    print("LG_CROSS_QK_GAIN =", tuple(cg))
    sd = {k: v.double() for k, v in W.make_lightglue_weights(1, self_qk_gain=sg, cross_qk_gain=cg).items()}
    e0, e1 = L.posenc(sd, k0), L.posenc(sd, k1)
    x0, x1 = d0, d1
    for i in range(9):
        lg, sim, y0, y1 = layer_stats(sd, i, x0, x1, e0, e1)
        print(f"layer {i}: |x| {x0.norm(dim=-1).mean():.2f}  self std {lg.std():.2f} max-prob {F.softmax(lg, -1).max(-1).values.mean():.3f}"
              f"  cross std {sim.std():.2f} max-prob {F.softmax(sim, -1).max(-1).values.mean():.3f}")
        x0, x1 = L.cross_block(sd, i, y0, y1)


if __name__ == "__main__":
    main()
