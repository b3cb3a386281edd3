"""GPU (-m gpu): multi-rank rehearsal on ONE GPU (VERDICT r02 "do this" 6).

No 8-GPU node was available to rounds 1-2, so the N > 1 code path (block sharding, fixed-stride padding, the all-gather, the
max-over-ranks timing, rank-0 reporting) had only ever executed on gloo/CPU tensors and at world size 1 on RCCL.  Here
`torch.distributed.run --nproc-per-node 2` launches the real scripts with BOTH ranks pinned to device 0
(SUPERSLAM_HIP_DEVICE=0): first on RCCL ("nccl"); RCCL refuses two ranks of one communicator on the same device ("Duplicate
GPU"), in which case the same launch is repeated on gloo (SUPERSLAM_DIST_BACKEND=gloo: device tensors staged through the host).
Either way every line of the N > 1 path runs against real device tensors before the first 8-GPU lease:
  * scripts/offline_extract.py (BASELINE config 3): the gathered (desc, kp, n) of 16 frames == a single-process run, bit for bit;
  * bench.py --gpus 2: one JSON line, n_gpus = 2, value = both ranks' pairs / the slower rank's time.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(script_args, backend, port, timeout=420, plain=False):
    """plain = the form the driver uses for one GPU, `python script.py --gpus 2 ...`, with NO launcher around it: the script
    re-executes itself under torch.distributed.run (superslam_amd/shard.py::relaunch_under_launcher_if_needed)."""
    env = dict(os.environ, SUPERSLAM_HIP_DEVICE="0", SUPERSLAM_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("LOCAL_RANK", None); env.pop("RANK", None); env.pop("WORLD_SIZE", None); env.pop("MASTER_PORT", None)
    cmd = [sys.executable] + script_args if plain else \
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
         "--master-port", str(port)] + script_args
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env, cwd=ROOT)
        return r.returncode, r.stdout, r.stderr
    except subprocess.TimeoutExpired as e:
        return 124, (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""), "timeout"


def _two_ranks(script_args, port, plain=False):
    """RCCL first; gloo when RCCL will not put two ranks on one device.  Returns (backend used, stdout)."""
    rc, out, err = _launch(script_args, "nccl", port, timeout=240, plain=plain)
    if rc == 0:
        return "nccl", out
    print(f"nccl with two ranks on device 0: rc {rc}; tail: {err[-600:]}")
    rc, out, err = _launch(script_args, "gloo", port + 1, plain=plain)
    assert rc == 0, (rc, out[-2000:], err[-3000:])
    return "gloo", out


def _last_json(text):
    for line in reversed(text.strip().splitlines()):
        line = line.strip()
        if line.startswith("{"):
            return json.loads(line)
    raise AssertionError("no JSON line in:\n" + text[-2000:])


def test_offline_extract_two_ranks_equals_single_process(tmp_path, parity_report):
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    args = ["--frames", "16", "--batch", "8", "--h", "240", "--w", "376", "--max-kp", "200"]
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "offline_extract.py"), *args, "--dump", one],
                       capture_output=True, text=True, timeout=300, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")})
    assert r.returncode == 0, r.stdout + r.stderr[-2000:]
    backend, out = _two_ranks([os.path.join(ROOT, "scripts", "offline_extract.py"), *args, "--dump", two, "--gpus", "2"], 29611, plain=True)
    j = _last_json(out)
    assert j["ranks"] == 2 and j["frames"] == 16 and j["backend"] == backend
    a, b = np.load(one), np.load(two)
    np.testing.assert_array_equal(a["n"], b["n"])
    np.testing.assert_array_equal(a["kp"].view(np.uint32), b["kp"].view(np.uint32))
    np.testing.assert_array_equal(a["desc"].view(np.uint16), b["desc"].view(np.uint16))
    assert int(a["n"].min()) > 0
    print(f"offline_extract, 2 ranks on one GPU ({backend}): gathered tensors == single-process run bit for bit; {j}")
    parity_report["multirank_rehearsal_offline_extract"] = {"backend": backend, "frames": 16, "bit_identical": True}


def test_bench_two_ranks_prints_one_line_with_n_gpus_2(parity_report):
    # the PLAIN form (no launcher): what the driver would run for N > 1 if it called bench.py the way it does for N = 1
    backend, out = _two_ranks([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--pairs", "8",
                               "--chunks", "2", "--headline-only"], 29631, plain=True)
    lines = [l for l in out.strip().splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, lines            # rank 0 alone reports
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["config"]["pairs_per_step"] == 16 and j["config"]["keypoints_found"] == [600, 600]
    # whole-job aggregate: both ranks' pairs over the slower rank's time
    assert abs(j["value"] - 2 * 16 * 2 / (j["ms_per_step"] * 2 / 1e3)) / j["value"] < 1e-3
    print(f"bench.py --gpus 2 on one GPU ({backend}): {j['value']} pairs/s, {j['ms_per_step']} ms/step")
    parity_report["multirank_rehearsal_bench"] = {"backend": backend, "value": j["value"], "n_gpus": j["n_gpus"]}


def test_rccl_exchange_through_the_c_abi_at_world_size_1(parity_report):
    """sship_comm_* / sship_gather_features_rccl (include/sship.h): the exchange step a C++ host of SuperSLAM calls.  One GPU here, so
    world = 1: the communicator initialises on RCCL, the grouped all-gather runs on the caller's stream and reproduces its input;
    the same call with world = 8 is what configs 3 / 5 use.  Also: torch.distributed's RCCL and the library's run-time-bound RCCL are
    the same loaded library (one copy per process)."""
    import torch

    from superslam_amd import _lib
    from superslam_amd.shard import RcclComm

    _lib.init(0)
    comm = RcclComm(0, 1)
    L = _lib.lib()
    assert L.sship_comm_rank(comm._h) == 0 and L.sship_comm_world(comm._h) == 1
    g = torch.Generator(device="cpu").manual_seed(5)
    desc = torch.randn((5, 200, 256), generator=g).half().cuda()
    kp = torch.rand((5, 200, 3), generator=g).cuda()
    n = torch.randint(0, 201, (5,), generator=g, dtype=torch.int32).cuda()
    da, ka, na = comm.gather_features(desc, kp, n)
    torch.cuda.synchronize()
    assert torch.equal(da, desc) and torch.equal(ka, kp) and torch.equal(na, n)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        da2, ka2, na2 = comm.gather_features(desc, kp, n, stream=s.cuda_stream)
    s.synchronize()
    assert torch.equal(da2, desc) and torch.equal(na2, n)
    comm.close()
    maps = open("/proc/self/maps").read()
    rccls = {l.split()[-1] for l in maps.splitlines() if "librccl" in l}
    assert len(rccls) == 1, rccls
    parity_report["rccl_c_abi_world1"] = {"ok": True, "rccl": sorted(rccls)[0]}
