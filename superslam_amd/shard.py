"""Multi-GPU scale-out of the front-end: one process per GPU, frames/pairs sharded, ONE all-gather.

The reference is single-GPU (SURVEY.md 2 'Parallelism': none).  Independent frames / stereo pairs carry
no cross-frame state, so they shard embarrassingly: rank r processes its block of units with replicated
weights and the only exchange is an all-gather of the fixed-stride padded results
  desc [units, max_kp, 256] f16, kp [units, max_kp, 3] f32, n [units] i32
into the shared host-side descriptor pool image (BASELINE config 3 / 5).  torch.distributed's "nccl"
backend is RCCL on ROCm (xGMI between the 8 GPUs of a node); the same code runs on "gloo" CPU tensors,
which is how it is tested without GPUs.
"""
from __future__ import annotations

from typing import List, Tuple


def shard_block(total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [start, stop) of `total` units for `rank`; the first total % world ranks get one extra."""
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_round_robin(total: int, rank: int, world: int) -> List[int]:
    """Unit ids u with u % world == rank (config 5: camera c -> rank c)."""
    return list(range(rank, total, world))


def pair_schedule(n_cameras: int, world: int) -> List[List[Tuple[int, int]]]:
    """Cross-camera LightGlue pairs (i < j) balanced over ranks (config 5: 28 pairs / 8 GPUs, greedy)."""
    pairs = [(i, j) for i in range(n_cameras) for j in range(i + 1, n_cameras)]
    out: List[List[Tuple[int, int]]] = [[] for _ in range(world)]
    for k, p in enumerate(pairs):
        out[k % world].append(p)
    return out


def dist_env():
    """(rank, device index, world size, under a launcher?, backend) of this process.

    One process per GPU: the device is LOCAL_RANK unless SUPERSLAM_HIP_DEVICE pins it (the single-GPU multi-rank rehearsal runs
    every rank on device 0).  The backend is RCCL ("nccl") unless SUPERSLAM_DIST_BACKEND says otherwise - RCCL refuses two ranks
    on one device, so that rehearsal falls back to "gloo" (collectives staged through the host) and still executes the same
    sharding / gather / timing code the 8-GPU run does."""
    import os

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dev = int(os.environ.get("SUPERSLAM_HIP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    return rank, dev, world, "RANK" in os.environ, os.environ.get("SUPERSLAM_DIST_BACKEND", "nccl")


def init_process_group(backend: str, device_index: int):
    import os

    import torch
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", device_index))
    else:
        dist.init_process_group(backend)


def all_reduce_max_seconds(dt: float) -> float:
    """max over ranks of a wall-clock interval (the slowest rank defines the step time)."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([dt], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_gather_features(desc, kp, n, total_units: int, group=None):
    """Gather per-rank padded results into global tensors ordered by unit id (block sharding).

    desc [u_local, K, 256], kp [u_local, K, 3], n [u_local]; ranks may own different unit counts (the tail
    ranks own one less), so each rank pads to ceil(total/world) units and the pad rows are dropped after the
    collective - a fixed-stride all-gather instead of an all-gather-v.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    per = (total_units + world - 1) // world
    outs = []
    via_host = dist.get_backend(group) != "nccl" and desc.is_cuda   # gloo: device tensors are staged through the host
    home = desc.device
    for t in (desc, kp, n):
        if via_host:
            t = t.cpu()
        pad = torch.zeros((per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        full = torch.empty((world * per,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        try:
            dist.all_gather_into_tensor(full, pad, group=group)
        except (RuntimeError, NotImplementedError):
            parts = [torch.empty_like(pad) for _ in range(world)]
            dist.all_gather(parts, pad, group=group)
            full = torch.cat(parts, 0)
        keep = []
        for r in range(world):
            a, b = shard_block(total_units, r, world)
            keep.append(full[r * per: r * per + (b - a)])
        outs.append(torch.cat(keep, 0).to(home) if via_host else torch.cat(keep, 0))
    return tuple(outs)


class HostDescriptorPool:
    """Host image of the gathered descriptors (the shared DescriptorPool of config 3), indexed by unit id."""

    def __init__(self, desc, kp, n):
        self.desc = desc.cpu()
        self.kp = kp.cpu()
        self.n = n.cpu()

    def features(self, unit: int):
        k = int(self.n[unit])
        return self.kp[unit, :k], self.desc[unit, :k]
