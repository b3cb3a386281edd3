#!/usr/bin/env python3
"""Developer A/B of the LightGlue call on the GPU box: whole-call time (device events, median of `reps` calls of `pairs` pairs) and the
isolated stage times of sship_lg_bench_stage, for whatever kernel selection the environment asks for
(SUPERSLAM_HIP_FFN, SUPERSLAM_HIP_LG_SPLIT, ...).  Also prints a checksum of matches0 / mscores0 and, with --ref, the agreement with a
reference run saved by --save (two fp16 paths: the bars of tests/_lgcmp.py apply).

  python scripts/dev/lg_ab.py --pairs 64 --save /tmp/ref.npz
  SUPERSLAM_HIP_FFN=16 SUPERSLAM_HIP_LG_SPLIT=1 python scripts/dev/lg_ab.py --pairs 64 --ref /tmp/ref.npz
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--kp", type=int, default=600)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--save", default=None)
    ap.add_argument("--ref", default=None)
    ap.add_argument("--tag", default="")
    ap.add_argument("--zero-weights", action="store_true", help="all-zero weights and descriptors (DVFS experiment: operand toggling vs clock)")
    args = ap.parse_args()
    import numpy as np
    import torch

    sys.path.insert(0, os.path.join(ROOT, "scripts")); import _devlib; _devlib.use_dev_library()  # SSHIP_DEV_LIBRARY -> explicit set_library_path
    from superslam_amd import LightGlue, _lib
    from superslam_amd.weights import make_lightglue_weights, save_safetensors

    torch.cuda.set_device(0)
    _lib.init(0)
    L = _lib.lib()
    d = tempfile.mkdtemp()
    wts = make_lightglue_weights(1)
    if args.zero_weights:
        wts = {k: v * 0 for k, v in wts.items()}
    save_safetensors(wts, os.path.join(d, "lg.safetensors"))
    P, K = args.pairs, args.kp
    lg = LightGlue(os.path.join(d, "lg.safetensors"), 1376, 376, max_keypoints=K, max_pairs=P)
    assert lg.initialize(), lg.last_error
    g = torch.Generator().manual_seed(7)
    # set 1 = a noisy permuted copy of set 0, so that there are real matches
    kp0 = torch.rand((P, K, 3), generator=g) * torch.tensor([1376.0, 376.0, 1.0])
    d0 = torch.nn.functional.normalize(torch.randn((P, K, 256), generator=g), dim=-1)
    perm = torch.stack([torch.randperm(K, generator=g) for _ in range(P)])
    kp1 = torch.gather(kp0, 1, perm[:, :, None].expand(-1, -1, 3)) + torch.randn((P, K, 3), generator=g) * torch.tensor([2.0, 2.0, 0.0])
    d1 = torch.nn.functional.normalize(torch.gather(d0, 1, perm[:, :, None].expand(-1, -1, 256)) + 0.15 * torch.randn((P, K, 256), generator=g), dim=-1)
    kp = torch.stack([kp0, kp1], 1).reshape(2 * P, K, 3).contiguous().cuda()
    desc = torch.stack([d0, d1], 1).reshape(2 * P, K, 256).half().contiguous().cuda()
    if args.zero_weights:
        desc.zero_()
    n = torch.full((2 * P,), K, dtype=torch.int32).cuda()
    n[1] = K - 37
    m0 = torch.empty((P, K), dtype=torch.int32, device="cuda")
    s0 = torch.empty((P, K), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream

    def call():
        _lib.check(L.sship_lg_match_batch_device(lg._h, kp.data_ptr(), n.data_ptr(), desc.data_ptr(), P, m0.data_ptr(), s0.data_ptr(), st))

    for _ in range(3):
        call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); call(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    out = {"tag": args.tag, "env": {k: v for k, v in os.environ.items() if k.startswith("SUPERSLAM_HIP") or k.startswith("SSHIP_")},
           "pairs": P, "kp": K, "call_ms_median": round(ts[len(ts) // 2], 4), "call_ms_min": round(ts[0], 4)}
    names = ["proj0", "self_attn", "cross_attn", "ffn_self+proj", "ffn_cross+proj", "ffn_last+final", "assign1", "assign2"]
    stages = {}
    for i, nm in enumerate(names):
        ms = C.c_float(0)
        _lib.check(L.sship_lg_bench_stage(lg._h, i, 10, C.byref(ms)))
        stages[nm] = round(ms.value, 4)
    out["stage_ms"] = stages
    call(); torch.cuda.synchronize()
    m, s = m0.cpu().numpy(), s0.cpu().numpy()
    out["matches"] = int((m >= 0).sum())
    out["checksum"] = [int(np.int64(m).sum()), float(np.float64(s).sum())]
    if args.save:
        np.savez(args.save, m=m, s=s)
    if args.ref:
        r = np.load(args.ref)
        same = (s > 0) == (r["s"] > 0)
        out["vs_ref"] = {"agreement": float((m == r["m"]).mean()), "flips": int((~same).sum()), "rows": int(m.size),
                         "mscores_maxd": float(np.abs(s - r["s"])[same].max())}
    print(json.dumps(out), flush=True)
    lg.close()


if __name__ == "__main__":
    main()
