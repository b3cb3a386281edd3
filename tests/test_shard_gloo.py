"""CPU: the N>1 path (sharding + the one all-gather) on world_size-2 gloo."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from superslam_amd.shard import all_gather_features, all_reduce_max_seconds, dist_env, pair_schedule, shard_block, shard_round_robin


def _fake(unit, k=6):
    g = torch.Generator().manual_seed(1000 + unit)
    n = int(torch.randint(0, k + 1, (1,), generator=g))
    desc = torch.zeros((k, 256), dtype=torch.float16)
    desc[:n] = torch.randn((n, 256), generator=g).half()
    kp = torch.zeros((k, 3))
    kp[:n] = torch.rand((n, 3), generator=g)
    return desc, kp, n


def _worker(rank, world, total, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = shard_block(total, rank, world)
    local = [_fake(u) for u in range(a, b)]
    desc = torch.stack([x[0] for x in local]) if local else torch.zeros((0, 6, 256), dtype=torch.float16)
    kp = torch.stack([x[1] for x in local]) if local else torch.zeros((0, 6, 3))
    n = torch.tensor([x[2] for x in local], dtype=torch.int32)
    gd, gk, gn = all_gather_features(desc, kp, n, total)
    ok = True
    for u in range(total):
        d, k, nn = _fake(u)
        ok &= bool(torch.equal(gd[u], d) and torch.equal(gk[u], k) and int(gn[u]) == nn)
    ok &= all_reduce_max_seconds(1.0 + rank) == float(world)     # bench.py's max-over-ranks step time
    q.put((rank, ok, tuple(gd.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [7, 8])
def test_all_gather_equals_single_process_concatenation(total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + total
    procs = [ctx.Process(target=_worker, args=(r, 2, total, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res), res
    assert all(shape == (total, 6, 256) for _, _, shape in res)


def _scale_worker(rank, world, port, q):
    import sys

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench

    res = bench.scale_extras(torch, dist, rank, world, "gloo", 2.0 + rank, 1000)    # rank r took 2 + r seconds for 1000 pairs
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_scale_extras_on_two_gloo_ranks():
    """bench.py's N > 1 extras (VERDICT r04 "do this" 7b) on CPU: every rank reports its own rate, and on a backend that is not RCCL the C-ABI
    exchange is skipped WITH a reason instead of being attempted (its collectives exist only on RCCL)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400) + 31
    procs = [ctx.Process(target=_scale_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    for r in (0, 1):
        assert res[r]["per_rank_pairs_per_s"] == [500.0, 333.33] and "skipped" in res[r]["exchange"] and "error" not in res[r], res[r]


def test_sharding_covers_every_unit_once():
    for total in (0, 1, 7, 8, 4096):
        for world in (1, 2, 4, 8):
            blocks = [shard_block(total, r, world) for r in range(world)]
            assert blocks[0][0] == 0 and blocks[-1][1] == total
            assert all(blocks[i][1] == blocks[i + 1][0] for i in range(world - 1))
            rr = sorted(u for r in range(world) for u in shard_round_robin(total, r, world))
            assert rr == list(range(total))
    sched = pair_schedule(8, 8)
    flat = sorted(p for r in sched for p in r)
    assert len(flat) == 28 and len(set(flat)) == 28 and max(len(r) for r in sched) - min(len(r) for r in sched) <= 1


def test_dist_env_device_pin_and_backend_override(monkeypatch):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SUPERSLAM_HIP_DEVICE", "SUPERSLAM_DIST_BACKEND"):
        monkeypatch.delenv(k, raising=False)
    assert dist_env() == (0, 0, 1, False, "nccl")
    monkeypatch.setenv("RANK", "3"); monkeypatch.setenv("LOCAL_RANK", "3"); monkeypatch.setenv("WORLD_SIZE", "8")
    assert dist_env() == (3, 3, 8, True, "nccl")                 # one process per GPU: device = LOCAL_RANK, RCCL
    monkeypatch.setenv("SUPERSLAM_HIP_DEVICE", "0"); monkeypatch.setenv("SUPERSLAM_DIST_BACKEND", "gloo")
    assert dist_env() == (3, 0, 8, True, "gloo")                 # the one-GPU multi-rank rehearsal (tests/test_gpu_multirank_rehearsal.py)


# ---- `python script.py --gpus N` without a launcher (VERDICT r03 "do this" 3) -----------------------------------------
def test_launcher_command_line_is_the_drivers_form():
    import sys

    from superslam_amd.shard import launcher_command

    cmd = launcher_command("/x/bench.py", ["--gpus", "8", "--steps", "20", "--warmup", "5"], 8, port=29555)
    assert cmd == [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
                   "--master-port", "29555", "/x/bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"]
    free = launcher_command("/x/bench.py", [], 2)     # no port given: the launcher's own store picks and holds one (no bind-and-close probe)
    assert "--master-port" not in free and free[free.index("--standalone"):][:3] == ["--standalone", "--local-addr", "127.0.0.1"]
    assert free[-1] == "/x/bench.py"


def test_relaunch_only_when_no_launcher_is_around(monkeypatch):
    from superslam_amd import shard

    calls = []
    monkeypatch.setattr(os, "execvpe", lambda f, a, e: calls.append((f, a, e)))
    monkeypatch.delenv("RANK", raising=False)
    shard.relaunch_under_launcher_if_needed(1, "/x/bench.py", ["--gpus", "1"])
    assert not calls                                              # N = 1: plain process, as the driver runs it
    shard.relaunch_under_launcher_if_needed(8, "/x/bench.py", ["--gpus", "8", "--steps", "3"])
    assert len(calls) == 1
    f, a, e = calls[0]
    assert a[1:9] == ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--standalone", "--local-addr", "127.0.0.1"]
    assert a[-5:] == ["/x/bench.py", "--gpus", "8", "--steps", "3"]
    assert e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("RANK", "3")
    shard.relaunch_under_launcher_if_needed(8, "/x/bench.py", ["--gpus", "8"])
    assert len(calls) == 1                                        # already one of the launcher's ranks: no second exec


def test_plain_invocation_with_gpus_2_really_runs_two_ranks(tmp_path):
    """End to end on CPU: a script that uses the same three helpers as bench.py (relaunch, dist_env, init_process_group), started as
    `python script.py --gpus 2` with no launcher, comes out as two gloo ranks of which rank 0 prints one line."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "two.py"
    script.write_text(f"""
import os, sys
sys.path.insert(0, {root!r})
from superslam_amd.shard import relaunch_under_launcher_if_needed, dist_env, init_process_group, all_reduce_max_seconds
relaunch_under_launcher_if_needed(2, os.path.abspath(__file__), sys.argv[1:])
import torch.distributed as dist
rank, dev, world, under, backend = dist_env()
init_process_group(backend, dev)
t = all_reduce_max_seconds(1.0 + rank)
if rank == 0:
    print("RESULT", world, under, t, sys.argv[1:], flush=True)
dist.barrier(); dist.destroy_process_group()
""")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["SUPERSLAM_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, str(script), "--gpus", "2"], capture_output=True, text=True, timeout=240, env=env)
    assert r.returncode == 0, r.stdout + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    assert lines == ["RESULT 2 True 2.0 ['--gpus', '2']"], (lines, r.stderr[-2000:])
