#!/usr/bin/env python3
"""BASELINE config 5: N-camera rig, one (block of) camera stream(s) per rank; per tick SuperPoint on the local
cameras, ONE all-gather of the padded descriptors/keypoints (RCCL over xGMI), then every rank runs its share of the
C(N,2) cross-camera LightGlue pairs (greedy round-robin schedule).

  python scripts/multicam.py                                   # 1 GPU: 8 cameras, 28 pairs locally
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/multicam.py
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cameras", type=int, default=8)
    ap.add_argument("--h", type=int, default=720)
    ap.add_argument("--w", type=int, default=1280)
    ap.add_argument("--max-kp", type=int, default=1024)
    ap.add_argument("--ticks", type=int, default=20)
    ap.add_argument("--gpus", type=int, default=1, help="N > 1 without a launcher: re-executes under torch.distributed.run with N ranks")
    ap.add_argument("--dump", default=None, help="every rank writes <dump>.rank<r>.npz: its pair list, matches0 / mscores0 of the last tick and the "
                                                  "gathered (desc, kp, n) - tests/test_gpu_configs_at_size.py checks BASELINE.md 4 row 5 with it")
    args = ap.parse_args()
    from superslam_amd.shard import relaunch_under_launcher_if_needed
    relaunch_under_launcher_if_needed(args.gpus, os.path.abspath(__file__), sys.argv[1:])

    import numpy as np
    import torch
    import torch.distributed as dist

    from superslam_amd import LightGlue, SuperPoint, _lib
    from superslam_amd.shard import all_gather_features, dist_env, init_process_group, pair_schedule, shard_block
    from superslam_amd.synth import make_frame
    from superslam_amd.weights import make_lightglue_weights, make_superpoint_weights, save_safetensors

    rank, local, world, use_dist, backend = dist_env()
    torch.cuda.set_device(local)
    if use_dist:
        init_process_group(backend, local)
    _lib.init(local)
    wdir = tempfile.mkdtemp(prefix="sship_w")
    save_safetensors(make_superpoint_weights(0), os.path.join(wdir, "sp.safetensors"))
    save_safetensors(make_lightglue_weights(1), os.path.join(wdir, "lg.safetensors"))
    a, b = shard_block(args.cameras, rank, world)
    my_pairs = pair_schedule(args.cameras, world)[rank]
    sp = SuperPoint(os.path.join(wdir, "sp.safetensors"), args.max_kp, 0.005, 4, max_batch=max(1, b - a))
    lg = LightGlue(os.path.join(wdir, "lg.safetensors"), args.w, args.h, max_keypoints=args.max_kp, max_pairs=max(1, len(my_pairs)))
    assert sp.initialize() and lg.initialize(), (sp.last_error, lg.last_error)

    base = make_frame(args.h, args.w, 515)
    cams = torch.from_numpy(np.stack([np.roll(base, (0, 24 * c), axis=(0, 1)) for c in range(a, b)])).cuda()  # overlapping views
    k = args.max_kp
    idx = torch.tensor([c for p in my_pairs for c in p], dtype=torch.long, device="cuda")

    def tick():
        desc, kp, n = sp.extract_batch_device(cams)
        if use_dist:
            gd, gk, gn = all_gather_features(desc, kp, n, args.cameras)
        else:
            gd, gk, gn = desc, kp, n
        if len(my_pairs) == 0:
            return None
        pd, pk, pn = gd.index_select(0, idx).contiguous(), gk.index_select(0, idx).contiguous(), gn.index_select(0, idx).contiguous()
        return lg.match_batch_device(pk, pn, pd)

    out = tick(); torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.ticks):
        out = tick()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if args.dump:
        desc, kp, n = sp.extract_batch_device(cams)
        gd, gk, gn = all_gather_features(desc, kp, n, args.cameras) if use_dist else (desc, kp, n)
        torch.cuda.synchronize()
        np.savez(f"{args.dump}.rank{rank}.npz", pairs=np.asarray(my_pairs, np.int32).reshape(-1, 2),
                 m0=out[0].cpu().numpy() if out is not None else np.zeros((0, k), np.int32),
                 ms0=out[1].cpu().numpy() if out is not None else np.zeros((0, k), np.float32),
                 desc=gd.cpu().numpy(), kp=gk.cpu().numpy(), n=gn.cpu().numpy())
    if rank == 0:
        m0 = out[0] if out is not None else None
        print(json.dumps({"cameras": args.cameras, "ranks": world, "pairs_total": args.cameras * (args.cameras - 1) // 2,
                          "pairs_this_rank": len(my_pairs), "ticks_per_s": round(args.ticks / dt, 2),
                          "matches_pair0": int((m0[0] >= 0).sum()) if m0 is not None else 0}), flush=True)
    sp.close(); lg.close()
    if use_dist:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
