#!/usr/bin/env python3
"""Generate the committed golden fixtures (run in the BUILD container only).

SuperPoint vectors come from an *import of the reference's own model definition*
(/root/reference/utils/convert_superpoint_to_onnx.py: SuperPoint + DenseSuperPoint), run on CPU with
the seeded synthetic weights of superslam_amd.weights (the real weights are missing blobs).  The
reference never travels to the GPU box; only these small input/output arrays do.

LightGlue vectors are SELF-CONSISTENCY vectors from oracle/lightglue_ref.py in fp64 (the upstream
package is absent: parity unpinned, see that file's header).

Host-path tables (keypoint normalisation, -1 filtering, select/top-k on adversarial maps) are
produced by oracle/hostpath_ref.c and cross-checked in the tests by an independent pure-Python
transcription of the cited reference lines.
"""
import importlib.util
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import hostpath as H, lightglue_ref as L, superpoint_ref as R  # noqa: E402
from superslam_amd.synth import make_frame  # noqa: E402
from superslam_amd.weights import (make_lightglue_weights, make_superpoint_weights,  # noqa: E402
                                   state_dict_sha256)

OUT = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/utils/convert_superpoint_to_onnx.py"


def main():
    torch.set_num_threads(1)  # fixed reduction order for reproducible fixtures
    spec = importlib.util.spec_from_file_location("ref_sp_export", REF)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    sd = make_superpoint_weights(0)
    lg = make_lightglue_weights(1)
    meta = {"sp_seed": 0, "lg_seed": 1, "sp_sha256": state_dict_sha256(sd), "lg_sha256": state_dict_sha256(lg),
            "sp_probe": {k: sd[k].flatten()[:4].tolist() for k in ("conv1a.weight", "convPb.weight", "convDb.bias")},
            "torch": torch.__version__, "numpy": np.__version__}

    model = ref.SuperPoint()
    model.load_state_dict(sd)
    dense = ref.DenseSuperPoint(model.eval(), 4).eval()

    # ---- G2/G3: reference SuperPoint on three small procedural frames (incl. odd width 249) ----
    for name, (h, w), seed in (("sp_64x64", (64, 64), 11), ("sp_120x160", (120, 160), 12),
                               ("sp_96x249", (96, 249), 13)):
        img = make_frame(h, w, seed, n_rects=24)
        x = R.preprocess_u8(torch.from_numpy(img)[None])
        with torch.no_grad():
            s_ref, d_ref = dense(x)                      # <- the reference's own forward
            feat = model.encode(x)
            logits_ref = model.convPb(model.relu(model.convPa(feat)))
        s_ref = s_ref[0].numpy(); d_ref = d_ref[0].numpy()
        sel = {}
        for mk in (16, 600):
            r = H.select_topk(s_ref, h, w, 0.005, 4, mk, d_ref.shape[1], d_ref.shape[2])
            g = H.gather_normalize(d_ref.astype(np.float16), r["cell_h"], r["cell_w"])
            sel[f"kp_{mk}"] = r["kp"]; sel[f"hw_{mk}"] = r["hw"]
            sel[f"cell_h_{mk}"] = r["cell_h"]; sel[f"cell_w_{mk}"] = r["cell_w"]
            sel[f"gathered_{mk}"] = g
        np.savez_compressed(os.path.join(OUT, name + ".npz"), image=img, logits=logits_ref[0].numpy(),
                            scores=s_ref, descriptors=d_ref.astype(np.float16), **sel)

    # ---- G4: adversarial score maps for select/top-k (ties, plateaus, borders, == thr) ----
    rng = np.random.Generator(np.random.PCG64(77))
    cases = {}
    a = np.zeros((40, 56), np.float32)
    a[10, 10] = a[10, 30] = a[20, 10] = a[30, 50] = 0.5          # exact ties -> larger h, then larger w first
    a[4, 4] = 0.9; a[3, 20] = 0.95; a[35, 51] = 0.8; a[36, 30] = 0.99; a[20, 52] = 0.7   # border = 4 edges
    a[15, 15] = np.float32(0.005); a[15, 16] = np.nextafter(np.float32(0.005), np.float32(1))  # thr is double 0.005
    a[25, 20:24] = 0.25                                          # plateau
    cases["ties"] = a
    b = rng.random((48, 64), dtype=np.float32)
    b[rng.random((48, 64)) < 0.9] = 0
    b = np.round(b * 16) / 16                                     # heavy quantisation -> many ties
    cases["quantised"] = b.astype(np.float32)
    cases["empty"] = np.zeros((24, 32), np.float32)
    c = rng.random((32, 40), dtype=np.float32) * 0.004            # everything below thr
    cases["below_thr"] = c
    gold = {}
    for k, m in cases.items():
        for mk in (5, 1000):
            r = H.select_topk(m, m.shape[0], m.shape[1] + 1, 0.005, 4, mk, m.shape[0] // 8, m.shape[1] // 8)
            gold[f"{k}_map"] = m
            gold[f"{k}_kp_{mk}"] = r["kp"]; gold[f"{k}_hw_{mk}"] = r["hw"]
            gold[f"{k}_cell_h_{mk}"] = r["cell_h"]; gold[f"{k}_cell_w_{mk}"] = r["cell_w"]
    np.savez_compressed(os.path.join(OUT, "select_cases.npz"), **gold)

    # ---- G5: LightGlue self-consistency vectors (fp64 oracle) ----
    lgv = {}
    for n0, n1, seed in ((7, 5, 21), (64, 64, 22), (97, 130, 23)):
        g = torch.Generator().manual_seed(seed)
        k0 = (torch.rand((1, n0, 2), generator=g) * 2 - 1) * torch.tensor([1.0, 0.27])
        d0 = torch.nn.functional.normalize(torch.randn((1, n0, 256), generator=g), dim=-1)
        perm = torch.randperm(max(n0, n1), generator=g)[:n1] % n0
        k1 = k0[:, perm] + 0.01 * torch.randn((1, n1, 2), generator=g)
        d1 = torch.nn.functional.normalize(d0[:, perm] + 0.15 * torch.randn((1, n1, 256), generator=g), dim=-1)
        d0h, d1h = d0.half(), d1.half()   # the boundary hands fp16 descriptors to the matcher
        with torch.no_grad():
            m0, ms0, it = L.match(lg, k0, d0h.float(), k1, d1h.float(), return_internals=True)
        tag = f"n{n0}x{n1}"
        lgv[tag + "_kpts0"] = k0[0].numpy(); lgv[tag + "_kpts1"] = k1[0].numpy()
        lgv[tag + "_desc0"] = d0h[0].numpy(); lgv[tag + "_desc1"] = d1h[0].numpy()
        lgv[tag + "_matches0"] = m0[0].numpy(); lgv[tag + "_mscores0"] = ms0[0].numpy()
        lgv[tag + "_sim"] = it["sim"][0].float().numpy()
        for layer in (0, 8):   # residual streams after the first and the last layer (both images)
            lgv[f"{tag}_x0_l{layer}"] = it["x0_layers"][layer][0].float().numpy()
            lgv[f"{tag}_x1_l{layer}"] = it["x1_layers"][layer][0].float().numpy()
    np.savez_compressed(os.path.join(OUT, "lightglue_selfcheck.npz"), **lgv)

    # ---- G6: host pre/post known-answer tables (hand-checkable) ----
    kp = np.array([[0, 0], [620.5, 188], [1241, 376], [100.25, 50.75], [1240, 0]], np.float32)
    meta["normalize_kpts"] = {"image_w": 1241, "image_h": 376, "kp": kp.tolist(),
                              "expected": H.normalize_kpts(kp, 1241, 376).tolist()}
    m0 = np.array([3, -1, 0, -1, 7, 2], np.int32); ms = np.array([0.9, 0.5, 0.25, 0.0, 1.0, 0.125], np.float32)
    q, t, d = H.filter_matches(m0, ms)
    meta["filter_matches"] = {"matches0": m0.tolist(), "mscores0": ms.tolist(), "query": q.tolist(),
                              "train": t.tolist(), "distance": d.tolist()}
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print("golden fixtures written to", OUT)
    for fn in sorted(os.listdir(OUT)):
        print(f"  {fn:32s} {os.path.getsize(os.path.join(OUT, fn)):9d} B")


if __name__ == "__main__":
    main()
