#!/usr/bin/env python3
"""Developer aid: joules per launch of isolated stages under the power poller (scripts/power_telemetry.py), for the shipped library or an A/B
build of it - the unit every kernel experiment of round 6 is priced in (the call runs on the board's power cap: time = energy / cap).

  python scripts/dev/stage_energy.py [--library lib/variants/x.so] [--sp 5,6,8] [--lg 1,2,3,4] [--seconds 1.0] [--tag name] [--kp 600]
SuperPoint layer ids (sship_sp_bench_layer): 1 conv1a+1b+pool, 2 conv2a, 3 conv2b+pool, 4 conv3a, 5 conv3b+pool, 6 conv4a, 7 conv4b, 8 convPa,
9 convPb, 12 k_nms_tile, 13 k_topk, 14 descriptor head, 15 conv2a + conv2b + pool fused (conv_fuse2.hip).  LightGlue stage ids (sship_lg_bench_stage): 0 first Wqkv, 1 self attention, 2 cross
attention, 3 SelfBlock FFN (+ to_qk | to_v), 4 CrossBlock FFN (+ Wqkv), 5 last FFN + final_proj, 6 / 7 assignment passes.
Prints one JSON line: {tag, library, rows: [{stage, launch_us, avg_W, sclk_MHz, joules_per_launch, samples}]}.
"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "scripts"))
SP_NAMES = {1: "conv1a+conv1b+pool", 2: "conv2a", 3: "conv2b+pool", 4: "conv3a", 5: "conv3b+pool", 6: "conv4a", 7: "conv4b", 8: "convPa", 9: "convPb",
            12: "k_nms_tile", 13: "k_topk", 14: "k_desc_head_sparse", 15: "conv2a+conv2b+pool (fused)"}
LG_NAMES = {0: "lg_wqkv0", 1: "lg_self_attention", 2: "lg_cross_attention", 3: "lg_self_ffn", 4: "lg_cross_ffn", 5: "lg_last_ffn", 6: "lg_assign1", 7: "lg_assign2"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--library", default=None)
    ap.add_argument("--sp", default="")
    ap.add_argument("--lg", default="")
    ap.add_argument("--calls", default="", help="comma list of: lg (one LightGlue call of --pairs pairs), fe (one whole front-end call)")
    ap.add_argument("--seconds", type=float, default=1.0)
    ap.add_argument("--tag", default="base")
    ap.add_argument("--pairs", type=int, default=64)
    ap.add_argument("--kp", type=int, default=600)
    a = ap.parse_args()
    import numpy as np
    import torch

    import power_telemetry
    from superslam_amd import _lib
    _lib.set_library_path(a.library)
    from superslam_amd import LightGlue, SuperPoint
    from superslam_amd.synth import make_stereo_pair
    from superslam_amd.weights import make_lightglue_weights, make_superpoint_weights, save_safetensors

    P, H, W, K = a.pairs, 376, 1376, a.kp
    torch.cuda.set_device(0); _lib.init(0); L = _lib.lib()
    d = tempfile.mkdtemp()
    save_safetensors(make_superpoint_weights(0), d + "/sp.safetensors"); save_safetensors(make_lightglue_weights(1), d + "/lg.safetensors")
    sp = SuperPoint(d + "/sp.safetensors", K, 0.005, 4, max_batch=2 * P); assert sp.initialize(), sp.last_error
    lg = LightGlue(d + "/lg.safetensors", W, H, max_keypoints=K, max_pairs=P); assert lg.initialize(), lg.last_error
    pairs = [make_stereo_pair(H, W, 1234 + i) for i in range(min(P, 8))]
    imgs = torch.from_numpy(np.stack([im for i in range(P) for im in pairs[i % len(pairs)]])).cuda()
    imgs = torch.stack([torch.roll(imgs[i], (i // 16) * 41, 0) for i in range(2 * P)])
    L.sship_set_profiling(1); desc, kp, n = sp.extract_batch_device(imgs); torch.cuda.synchronize(); L.sship_set_profiling(0)
    lg.match_batch_device(kp, n, desc); torch.cuda.synchronize()
    ms = C.c_float(0)
    rows = []
    with power_telemetry.PowerPoller(0, hz=50.0) as poller:
        assert poller.ok, "no power telemetry backend"

        def one(name, run):
            run(10)
            iters = max(10, int(a.seconds * 1e3 / max(ms.value, 1e-3)))
            ta = time.monotonic(); run(iters); tb = time.monotonic()
            w = poller.window(ta, tb, settle_s=0.25 * (tb - ta))
            rows.append({"stage": name, "launch_us": round(ms.value * 1e3, 2), "avg_W": w["avg_W"], "sclk_MHz": w["sclk_MHz"],
                         "joules_per_launch": round(w["avg_W"] * ms.value * 1e-3, 5), "samples": w["n"]})

        for lid in [int(x) for x in a.sp.split(",") if x]:
            one(SP_NAMES.get(lid, f"sp{lid}"), lambda it, lid=lid: _lib.check(L.sship_sp_bench_layer(sp._h, lid, 2 * P, H, W, it, C.byref(ms), None)))
        for sid in [int(x) for x in a.lg.split(",") if x]:
            one(LG_NAMES.get(sid, f"lg{sid}"), lambda it, sid=sid: _lib.check(L.sship_lg_bench_stage(lg._h, sid, it, C.byref(ms))))
        from superslam_amd import FrontEndBatch
        fe = FrontEndBatch(sp, lg, P, H, W) if "fe" in a.calls else None
        stream = torch.cuda.current_stream().cuda_stream

        def call_loop(fn):
            def run(it):
                torch.cuda.synchronize(); t = time.perf_counter()
                for _ in range(it):
                    fn()
                torch.cuda.synchronize(); ms.value = (time.perf_counter() - t) / it * 1e3
            return run
        for c in [x for x in a.calls.split(",") if x]:
            if c == "lg":
                one("lightglue_call", call_loop(lambda: lg.match_batch_device(kp, n, desc, stream=stream)))
            elif c == "fe":
                one("frontend_call", call_loop(lambda: fe.run(imgs, stream)))
    print(json.dumps({"tag": a.tag, "library": os.path.relpath(_lib.LIB_PATH, ROOT), "keypoints": K, "rows": rows}), flush=True)
    sp.close(); lg.close()


if __name__ == "__main__":
    main()
