"""GPU, needs TWO devices: the C-ABI exchange at world size 2 over RCCL (VERDICT r04 "do this" 7a).

sship_comm_create / sship_gather_features_rccl (include/sship.h; csrc/shard_rccl.hip) are what a C++ host of SuperSLAM calls to collect the
descriptors of sharded frames into the shared DescriptorPool image (BASELINE configs[2]).  On the one-GPU boxes this repository has
seen, RCCL refuses two ranks on one device, so the grouped three-tensor all-gather had only ever run at world size 1
(tests/test_gpu_multirank_rehearsal.py).  This test is the one-shot for the first multi-GPU lease: it SKIPS on a one-GPU box and proves the
exchange the first time two devices are visible - two processes, one device each, NO torch.distributed: the 128-byte id goes from rank 0 to
rank 1 through a file, as a C++ launcher would hand it over.  Buffers are configs[2]-sized (512 frames x 600 keypoints per rank: 157 MB of
descriptors per rank); every rank regenerates both ranks' seeded contents and requires gathered == concatenation, bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _device_count():
    try:
        import torch

        return torch.cuda.device_count()
    except Exception:
        return 0


_WORKER = r"""
import ctypes as C, os, sys, time
import numpy as np, torch
sys.path.insert(0, {root!r})
rank, world, idfile = int(sys.argv[1]), 2, sys.argv[2]
os.environ["SUPERSLAM_HIP_DEVICE"] = str(rank)
from superslam_amd import _lib
from superslam_amd.shard import RcclComm
torch.cuda.set_device(rank)
_lib.init(rank)
L = _lib.lib()
if rank == 0:
    b = C.create_string_buffer(128)
    _lib.check(L.sship_comm_unique_id(b))
    open(idfile + ".tmp", "wb").write(b.raw)
    os.replace(idfile + ".tmp", idfile)
    idb = b.raw
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 120, "rank 0 never published the communicator id"
        time.sleep(0.05)
    idb = open(idfile, "rb").read()
comm = RcclComm(rank, world, id_bytes=idb)
assert L.sship_comm_rank(comm._h) == rank and L.sship_comm_world(comm._h) == world
U, K = {units}, {kp}

def contents(r):
    g = torch.Generator().manual_seed(1000 + r)
    d = torch.randn((U, K, 256), generator=g).half()
    k = torch.rand((U, K, 3), generator=g)
    n = torch.randint(0, K + 1, (U,), generator=g, dtype=torch.int32)
    return d, k, n

d, k, n = (t.cuda() for t in contents(rank))
before = torch.cuda.current_device()
da, ka, na = comm.gather_features(d, k, n)
torch.cuda.synchronize()
assert torch.cuda.current_device() == before          # the call leaves the caller's device binding alone (ADVICE r04)
for r in range(world):
    dr, kr, nr = contents(r)
    assert torch.equal(da[r * U:(r + 1) * U].cpu(), dr), ("desc", r)
    assert torch.equal(ka[r * U:(r + 1) * U].cpu(), kr), ("kp", r)
    assert torch.equal(na[r * U:(r + 1) * U].cpu(), nr), ("n", r)
# timing of the exchange itself: 5 calls, device events
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5):
    comm.gather_features(d, k, n)
e1.record(); e1.synchronize()
ms = e0.elapsed_time(e1) / 5
per_rank = U * K * (512 + 12) + 4 * U
print("RCCL_WORLD2_OK rank %d: %d units x %d keypoints, %.1f MB per rank, %.3f ms per gather, %.1f GB/s received" %
      (rank, U, K, per_rank / 1e6, ms, per_rank * (world - 1) / ms / 1e6), flush=True)
comm.close()
"""


@pytest.mark.skipif(_device_count() < 2, reason="needs two GPUs: RCCL refuses two ranks of one communicator on the same device")
@pytest.mark.parametrize("units,kp", [(512, 600), (1, 1024)])   # BASELINE configs[2] (157 MB per rank) and configs[4] (0.5 MB per rank)
def test_c_abi_exchange_over_rccl_at_world_size_2(tmp_path, units, kp):
    idfile = str(tmp_path / "rccl_id.bin")
    code = _WORKER.format(root=ROOT, units=units, kp=kp)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "SUPERSLAM_HIP_DEVICE"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), idfile], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for r, (rc, o, e) in enumerate(outs):
        print(o.strip())
        assert rc == 0 and "RCCL_WORLD2_OK" in o, (r, rc, o[-1500:], e[-3000:])
