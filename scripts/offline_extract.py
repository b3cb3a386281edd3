#!/usr/bin/env python3
"""BASELINE config 3: batched offline SuperPoint extraction, frames sharded over the ranks of one node, ONE
all-gather (RCCL over xGMI; fixed-stride padded tensors) of descriptors / keypoints / counts into the shared host
descriptor-pool image.

  python scripts/offline_extract.py --frames 256                      # 1 GPU
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/offline_extract.py --frames 4096

Prints one JSON line: frames/s (extraction), all-gather GB/s, and - with --check on 1 rank - that the gathered
tensor equals the directly computed one.
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
    ap.add_argument("--frames", type=int, default=256)
    ap.add_argument("--h", type=int, default=480)
    ap.add_argument("--w", type=int, default=752)
    ap.add_argument("--max-kp", type=int, default=600)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--gpus", type=int, default=1, help="N > 1 without a launcher: re-executes under torch.distributed.run with N ranks")
    ap.add_argument("--dump", default=None, help="rank 0 writes the gathered (desc, kp, n) to this .npz (the multi-rank rehearsal test "
                                                  "compares it bit-for-bit with a single-process run)")
    args = ap.parse_args()
    from superslam_amd.shard import relaunch_under_launcher_if_needed
    relaunch_under_launcher_if_needed(args.gpus, os.path.abspath(__file__), sys.argv[1:])

    import numpy as np
    import torch
    import torch.distributed as dist

    from superslam_amd import SuperPoint, _lib
    from superslam_amd.shard import HostDescriptorPool, all_gather_features, dist_env, init_process_group, shard_block
    from superslam_amd.synth import make_frame
    from superslam_amd.weights import make_superpoint_weights, save_safetensors

    rank, local, world, use_dist, backend = dist_env()
    torch.cuda.set_device(local)
    if use_dist:
        init_process_group(backend, local)
    _lib.init(local)
    wdir = tempfile.mkdtemp(prefix="sship_w")
    save_safetensors(make_superpoint_weights(0), os.path.join(wdir, "sp.safetensors"))
    sp = SuperPoint(os.path.join(wdir, "sp.safetensors"), args.max_kp, 0.005, 4, max_batch=args.batch)
    assert sp.initialize(), sp.last_error

    a, b = shard_block(args.frames, rank, world)
    base = make_frame(args.h, args.w, 4242)
    def frame(f):   # cheap distinct procedural frames: a 2-D roll of one base image, keyed by the global frame id
        return np.roll(base, ((f * 37) % args.h, (f * 101) % args.w), axis=(0, 1))
    local_frames = torch.from_numpy(np.stack([frame(f) for f in range(a, b)])).cuda() if b > a else \
        torch.zeros((0, args.h, args.w), dtype=torch.uint8, device="cuda")
    k = args.max_kp
    desc = torch.zeros((b - a, k, 256), dtype=torch.float16, device="cuda")
    kp = torch.zeros((b - a, k, 3), dtype=torch.float32, device="cuda")
    n = torch.zeros((b - a,), dtype=torch.int32, device="cuda")

    def run():
        for i in range(0, b - a, args.batch):
            j = min(i + args.batch, b - a)
            sp.extract_batch_device(local_frames[i:j], desc[i:j], kp[i:j], n[i:j])
    run(); torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t0 = time.perf_counter()
    run(); torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    t_ext = time.perf_counter() - t0
    out = {"frames": args.frames, "ranks": world, "frames_per_s": round(args.frames / t_ext, 1), "backend": backend if use_dist else None}
    if use_dist:
        t0 = time.perf_counter()
        gd, gk, gn = all_gather_features(desc, kp, n, args.frames)
        torch.cuda.synchronize(); dist.barrier()
        t_ag = time.perf_counter() - t0
        out["allgather_GBps_per_rank_out"] = round(gd.numel() * 2 / t_ag / 1e9, 2)
    else:
        gd, gk, gn = desc, kp, n
    if rank == 0:
        pool = HostDescriptorPool(gd, gk, gn)        # the shared host descriptor-pool image
        kp0, d0 = pool.features(0)
        out["frame0_keypoints"] = int(kp0.shape[0]); out["pool_bytes"] = int(pool.desc.numel() * 2)
        if args.dump:
            np.savez(args.dump, desc=pool.desc.numpy(), kp=pool.kp.numpy(), n=pool.n.numpy())
        print(json.dumps(out), flush=True)
    sp.close()
    if use_dist:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
