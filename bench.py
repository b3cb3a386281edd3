#!/usr/bin/env python3
"""bench.py - stereo pairs/sec of the front-end hot path (SuperPoint x2 + select + gather + LightGlue)
on 1376x376 KITTI-shaped synthetic frames (BASELINE.json metric / configs[1]).

  python bench.py --gpus N --steps K --warmup W      (N > 1 without a launcher: re-executes itself as the next line)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A step is ONE pass of the hot path over one batch of synthetic input: `--chunks` x `--pairs` stereo pairs that are
already resident in HBM, processed as `--chunks` calls of sship_frontend_batch_device (the reference's unit of work per
pair: 1x extract_stereo + 1x device match, src/StereoFrontEnd.cc:14,33).  The default batch (8 x 64 = 512 pairs, every
chunk different images) makes 20 timed steps last > 2 s.  Independent pairs shard across ranks with no data-path
collective (weak scaling: every rank runs its own batch); `value` = all pairs of all ranks / max-over-ranks wall time.

Prints ONE JSON line with the driver's fields plus
  self_check    : after the timed region, one chunk's batched outputs against the per-pair path (keypoints bit-identical,
                  descriptors <= 1 ulp, matches within the matcher bars); the process exits non-zero if it fails
  roofline      : the dominant kernel (conv1a+conv1b+pool, 36 % of the pair's FLOPs); `traffic` = HBM bytes per launch measured IN THIS RUN
                  (two rocprofv3 --pmc passes as child processes; `traffic_source` says measured or, after a failure, read from the committed file);
                  `frac` from the IN-SITU launch duration
                  (HIP events around the launch inside profiled headline calls on the stream it runs on - the figure
                  rocprofv3 --kernel-trace of the same steps reproduces), `frac_isolated` from 20 back-to-back re-launches
  roofline_mfma : the same for every matrix-core stage (conv layers in-situ + isolated, LightGlue stages isolated)
  roofline_hbm  : achieved GB/s against 8 TB/s for the memory-bound stages (NMS tile kernel, convPb, descriptor head /
                  gather, assignment passes), algorithmic bytes stated per entry
  single_pair_protocol : BASELINE.md's protocol to the letter on the per-frame unit (50 warm-up + 500 timed pairs, one pair per call,
                  median and p95 from device events); latency_ms_single_pair = median of five 50-call blocks of the same unit
  n1024         : the same metric with 1024 keypoints per image (the reference engine's upper profile)
  end_to_end    : the same step with the u8 images uploaded from pinned host memory (double buffered) and keypoints /
                  matches copied back inside the timed region (PCIe-inclusive; never `value`)
  power         : socket watts / shader MHz sampled by a poller thread (scripts/power_telemetry.py, >= 20 Hz) DURING the timed steps:
                  {avg_W, cap_W, sclk_MHz, joules_per_pair, pairs_per_s_at_cap}; every matrix kernel of the call runs on the board's power
                  cap, so joules per pair is the unit kernel work is priced in; `joules_per_launch` on every roofline_mfma / roofline_hbm row
                  comes from a >= 0.6 s loop of that stage under the same poller (rank 0; --no-power switches it off)
  cpu_baseline  : the CPU oracle (kind "port") on this box's host cores, bounded sample, rank 0 at N = 1 only.
--headline-only runs the warm-up and the timed steps and nothing else (profiling passes: every launch is a headline launch).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

H, W = 376, 1376
MFMA_PEAK_TFLOPS = 2500.0   # dense fp16/bf16 MFMA, MI355X_MICROARCH.md (never the 2:1-sparse figure)
HBM_PEAK_GBS = 8000.0
PATH_VS_PATH_BAR = 3e-2     # |d mscores0| between two fp16 paths (tests/_lgcmp.py derives it; tests/test_parity_margins.py pins the two together)


def sp_flops_per_image(h, w):
    """2*MAC per image, layer table utils/convert_superpoint_to_onnx.py:38-49."""
    h2, w2, h4, w4, hc, wc = h // 2, w // 2, h // 4, w // 4, h // 8, w // 8
    mac = (h * w * (9 * 64 + 576 * 64) + h2 * w2 * (576 * 64 * 2) + h4 * w4 * (576 * 128 + 1152 * 128)
           + hc * wc * (1152 * 128 * 2 + 1152 * 256 * 2 + 256 * 65 + 256 * 256))
    return 2.0 * mac


def lg_flops_per_pair(n):
    mac = 9 * (2 * n * 1245184 + 1792 * n * n) + 2 * n * 65792 + 256 * n * n
    return 2.0 * mac


def executed_flops_per_pair(h, w, n):
    """FLOPs the library really executes per pair: the algorithmic count minus (a) the dense descriptor head (convDa + convDb on
    all Hc x Wc cells), which is evaluated only at the <= n selected keypoints (k_desc_head_sparse), and (b) out_proj / to_out of
    the 18 attention blocks, which are folded into ffn.0's weights on the host."""
    hc, wc = h // 8, w // 8
    dense_desc = 2.0 * hc * wc * (1152 * 256 + 256 * 256)
    sparse_desc = 2.0 * n * (1152 * 256 + 256 * 256)
    folded = 2.0 * 9 * 2 * (2 * n) * 256 * 256
    return 2 * (sp_flops_per_image(h, w) - dense_desc + sparse_desc) + lg_flops_per_pair(n) - folded


def usable_cores():
    """Host cores this process may really use: affinity mask and cgroup CPU quota, not the machine's core count
    (oversubscribed OpenMP teams spin and make the CPU leg take minutes)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return max(1, min(n, 64))


def cpu_baseline(spw, lgw, left, right, max_kp, budget_s=20.0):
    """The CPU oracle on the host cores: same synthetic pair, fp32, all cores."""
    import numpy as np
    import torch

    from oracle import hostpath as Hh
    from oracle import lightglue_ref as LR
    from oracle import superpoint_ref as R

    cores = usable_cores()
    torch.set_num_threads(cores)

    def one_pair():
        x = R.preprocess_u8(torch.from_numpy(np.stack([left, right])))
        with torch.no_grad():
            s, d = R.dense_forward(spw, x)
        d16 = d.half().numpy()
        feats = []
        for b in range(2):
            sel = Hh.select_topk(s[b].numpy(), H, W, 0.005, 4, max_kp, H // 8, W // 8)
            feats.append((sel["kp"], Hh.gather_normalize(d16[b], sel["cell_h"], sel["cell_w"])))
        k0 = torch.from_numpy(Hh.normalize_kpts(feats[0][0], W, H))[None]
        k1 = torch.from_numpy(Hh.normalize_kpts(feats[1][0], W, H))[None]
        with torch.no_grad():
            LR.match(lgw, k0, torch.from_numpy(feats[0][1].astype(np.float32))[None], k1,
                     torch.from_numpy(feats[1][1].astype(np.float32))[None], dtype=torch.float32)

    one_pair()  # warm-up (thread pools, allocator)
    times = []
    t_end = time.perf_counter() + budget_s
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 10):
        t0 = time.perf_counter()
        one_pair()
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(1.0 / med, 4), "unit": "pairs/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} timed 1376x376 pairs (median of {len(times)}, 1 warm-up), fp32 torch-CPU SuperPoint + "
                      f"C select/gather + fp32 LightGlue, {cores} threads"}


def self_check(torch, np, fe, chunk, stream, wdir, max_kp):
    """After the timed region: the batched call's outputs for one chunk against the PER-PAIR path (a second pair of handles sized
    like the reference's per-frame use: SuperPoint max_batch = 2, LightGlue max_pairs = 1 - different kernel variants: latency-mode
    attention / FFN, one workgroup round).  Keypoints, scores and counts must be bit-identical, descriptors within 1 fp16 ulp,
    matches0 / mscores0 within the matcher bars of tests/_lgcmp.py (>= 99 % of the rows, |d mscores0| <= PATH_VS_PATH_BAR = 3e-2
    where the mutual flag agrees: this compares two fp16 paths, each of which may sit 2e-2 from the oracle - the 2e-2 bar is the
    path-vs-ORACLE tolerance of SURVEY 8(c), and the tests that compare with the oracle keep it).  The headline number is only meaningful if the batch it times computes what the per-frame path computes."""
    from superslam_amd import FrontEndBatch, LightGlue, SuperPoint

    P = fe.pairs
    fe.run(chunk, stream); torch.cuda.synchronize()
    kp, n, desc = fe.kp.cpu().numpy(), fe.n.cpu().numpy(), fe.desc.cpu().numpy()
    m0, s0 = fe.matches0.cpu().numpy(), fe.mscores0.cpu().numpy()
    sp1 = SuperPoint(os.path.join(wdir, "sp.safetensors"), max_kp, 0.005, 4, max_batch=2)
    lg1 = LightGlue(os.path.join(wdir, "lg.safetensors"), W, H, max_keypoints=max_kp, max_pairs=1)
    assert sp1.initialize() and lg1.initialize()
    fe1 = FrontEndBatch(sp1, lg1, 1, H, W)
    res = {"pairs": P, "kp_bit_identical": 0, "desc_max_ulp": 0, "matches_rows": 0, "matches_equal_rows": 0, "mutual_flips": 0,
           "mscores_maxd": 0.0, "min_pair_agreement": 1.0, "matches_batch": int((m0 >= 0).sum()), "matches_per_pair_path": 0}
    for p in range(P):
        fe1.run(chunk[2 * p:2 * p + 2], stream); torch.cuda.synchronize()
        kp1, n1, d1 = fe1.kp.cpu().numpy(), fe1.n.cpu().numpy(), fe1.desc.cpu().numpy()
        ok = np.array_equal(n1, n[2 * p:2 * p + 2])
        for b in range(2):
            k = int(n1[b])
            ok = ok and np.array_equal(kp1[b, :k].view(np.uint32), kp[2 * p + b, :k].view(np.uint32))
            a16, b16 = (np.where(v < 0, -(v & 0x7fff), v) for v in      # sign-magnitude fp16 bits -> a monotonic integer line
                        (d1[b, :k].view(np.int16).astype(np.int32), desc[2 * p + b, :k].view(np.int16).astype(np.int32)))
            if ok and k:
                res["desc_max_ulp"] = max(res["desc_max_ulp"], int(np.abs(a16 - b16).max()))
        res["kp_bit_identical"] += int(ok)
        k0 = int(n1[0])
        ma, sa = m0[p, :k0], s0[p, :k0]
        mb, sb = fe1.matches0.cpu().numpy()[0, :k0], fe1.mscores0.cpu().numpy()[0, :k0]
        ds = np.abs(sa - sb)
        flagdiff = (sa > 0) != (sb > 0)     # tests/_lgcmp.py: flips are judged by their size, mscores_maxd is taken where the flag agrees
        same = ~flagdiff
        res["matches_rows"] += k0
        res["matches_equal_rows"] += int((ma == mb).sum())
        res["mutual_flips"] += int((flagdiff & (ds > PATH_VS_PATH_BAR)).sum())
        res["small_flips"] = res.get("small_flips", 0) + int((flagdiff & (ds <= PATH_VS_PATH_BAR)).sum())
        res["mscores_maxd"] = max(res["mscores_maxd"], float(ds[same].max()) if same.any() else 0.0)
        res["mscores_maxd_all"] = max(res.get("mscores_maxd_all", 0.0), float(ds.max()) if k0 else 0.0)   # incl. rows whose mutual flag flips (keep regressions visible)
        res["min_pair_agreement"] = min(res["min_pair_agreement"], float((ma == mb).mean()) if k0 else 1.0)
        res["matches_per_pair_path"] += int((mb >= 0).sum())
    sp1.close(); lg1.close()
    res["mscores_maxd"] = round(res["mscores_maxd"], 5)
    res["mscores_maxd_all"] = round(res.get("mscores_maxd_all", 0.0), 5)
    res["min_pair_agreement"] = round(res["min_pair_agreement"], 4)
    res["agreement"] = round(res["matches_equal_rows"] / max(1, res["matches_rows"]), 5)
    res["ok"] = bool(res["kp_bit_identical"] == P and res["desc_max_ulp"] <= 1 and res["agreement"] >= 0.99
                     and res["mutual_flips"] <= max(1, int(0.005 * res["matches_rows"])) and res["mscores_maxd"] <= PATH_VS_PATH_BAR)
    res["mscores_bar"] = PATH_VS_PATH_BAR
    res["mscores_margin"] = round(PATH_VS_PATH_BAR / max(res["mscores_maxd"], 1e-9), 3)
    return res


def scale_extras(torch, dist, rank, world, backend, dt_own, pairs_per_rank):
    """N > 1 only, every rank calls it (after the timed region): what ONE multi-GPU lease should answer besides the curve -
      per_rank_pairs_per_s : each rank's own rate over the timed steps (the headline divides by the slowest);
      exchange             : the one collective of the sharded front-end, sship_gather_features_rccl through the C ABI, at the
                             two sizes BASELINE.json names: configs[2] (512 frames x 600 keypoints per rank: 157 MB of descriptors per
                             rank) and configs[4] (one 1024-keypoint frame per rank: 0.5 MB), 2 warm-up + 5 timed calls each, device
                             events on the launch stream; GB/s = bytes a rank RECEIVES from its world - 1 peers / time, and the gathered
                             tensor is checked against every rank's own pattern;
      rccl_world           : the rank count RCCL itself reports for the communicator.
    Never fails the bench line: an error becomes a string in the record.  Every phase that contains a collective of the C-ABI communicator is
    preceded by a VOTE over torch.distributed (which the timed region has just used): if any rank failed the phase before, all ranks skip the
    rest together - a rank that raised must not leave the others waiting inside a collective."""
    res = {"backend": backend}
    dev = "cuda" if backend == "nccl" else "cpu"

    def all_ok(ok):
        t = torch.tensor([0 if ok else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item()) == 0

    try:
        own = [None] * world
        dist.all_gather_object(own, round(pairs_per_rank / dt_own, 2))
        res["per_rank_pairs_per_s"] = own
        if backend != "nccl":
            res["exchange"] = "skipped: RCCL only (this run's collectives go through " + backend + ")"
            return res
        import ctypes as C

        from superslam_amd import _lib
        from superslam_amd.shard import RcclComm

        L = _lib.lib()
        # phase 1: the communicator id on rank 0 (no collective), then a broadcast that every rank takes part in whatever happened
        box, err = [None], None
        if rank == 0:
            try:
                buf = C.create_string_buffer(128)
                _lib.check(L.sship_comm_unique_id(buf))
                box[0] = buf.raw
            except Exception as e:  # noqa: BLE001
                err = f"{type(e).__name__}: {e}"[:300]
        dist.broadcast_object_list(box, src=0)
        if box[0] is None:
            res["error"] = "rank 0 could not create an RCCL id" + (": " + err if err else "")
            return res
        # phase 2: communicator (a collective inside RCCL: every rank calls it; the vote afterwards catches a rank whose call returned an error)
        comm, err = None, None
        try:
            comm = RcclComm(rank, world, id_bytes=box[0])
        except Exception as e:  # noqa: BLE001
            err = f"{type(e).__name__}: {e}"[:300]
        if not all_ok(comm is not None):
            res["error"] = "sship_comm_create failed on at least one rank" + (": " + err if err else "")
            return res
        res["rccl_world"] = int(L.sship_comm_world(comm._h))
        ex = {}
        for name, units, kp in (("configs[2] offline extraction", 512, 600), ("configs[4] camera rig tick", 1, 1024)):
            bufs, err = None, None
            try:   # allocation only: no collective in here
                desc = torch.full((units, kp, 256), float(rank + 1), dtype=torch.float16, device="cuda")
                kpt = torch.full((units, kp, 3), float(rank) + 0.5, dtype=torch.float32, device="cuda")
                n = torch.full((units,), kp - rank, dtype=torch.int32, device="cuda")
                bufs = (desc, kpt, n)
            except Exception as e:  # noqa: BLE001
                err = f"{type(e).__name__}: {e}"[:300]
            if not all_ok(bufs is not None):
                ex[name] = "skipped: a rank could not allocate its buffers" + (": " + err if err else "")
                continue
            desc, kpt, n = bufs
            for _ in range(2):
                da, ka, na = comm.gather_features(desc, kpt, n)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                da, ka, na = comm.gather_features(desc, kpt, n)
            e1.record(); e1.synchronize()
            ms = e0.elapsed_time(e1) / 5
            ok = all(bool((da[r * units:(r + 1) * units] == float(r + 1)).all()) and bool((na[r * units:(r + 1) * units] == kp - r).all())
                     and bool((ka[r * units:(r + 1) * units] == float(r) + 0.5).all()) for r in range(world))
            per_rank = units * kp * (512 + 12) + units * 4
            t = torch.tensor([ms], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            okall = all_ok(ok)
            ex[name] = {"bytes_per_rank": per_rank, "ms": round(float(t.item()), 4), "gathered_equals_ranks_patterns_on_every_rank": okall,
                        "ingest_gb_per_s_per_rank": round(per_rank * (world - 1) / (float(t.item()) * 1e-3) / 1e9, 2),
                        "aggregate_gb_per_s": round(per_rank * (world - 1) * world / (float(t.item()) * 1e-3) / 1e9, 2)}
            del desc, kpt, n, da, ka, na, bufs
        res["exchange"] = ex
        comm.close()
    except Exception as e:  # noqa: BLE001 - the curve must survive a failing extra
        res["error"] = f"{type(e).__name__}: {e}"[:400]
    return res


def make_chunks(torch, base, chunks):
    """`chunks` different image sets [2P,H,W] u8 in HBM derived from the P generated pairs: chunk c is the base set rolled
    vertically by 41 c rows (same roll for left and right: the row-band disparities stay a valid stereo geometry) and
    intensity-inverted for odd c - every chunk has its own keypoints and matches without 0.1 s of host synthesis per pair."""
    out = []
    for c in range(chunks):
        x = torch.roll(base, shifts=41 * c, dims=1) if c else base
        out.append((255 - x) if c % 2 else x.clone())
    return out


def main():
    import faulthandler

    faulthandler.dump_traceback_later(int(os.environ.get("BENCH_WATCHDOG_S", "900")), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=64, help="stereo pairs per library call (per GPU)")
    ap.add_argument("--chunks", type=int, default=8, help="library calls per step: a step processes chunks x pairs pairs resident in HBM")
    ap.add_argument("--max-kp", type=int, default=600, help="superpoint.max_keypoints (600 = the KITTI YAML)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="warm-up + timed steps only (profiling passes)")
    ap.add_argument("--measure-traffic", dest="measure_traffic", action="store_true", default=True,
                    help="(default) measure roofline.traffic IN THIS RUN: two rocprofv3 --pmc passes over `bench.py --headline-only` as child processes "
                         "(scripts/pmc_traffic.sh, ~20-60 s, 150 s limit; needs rocprofv3 on PATH).  On any failure the figure is READ from "
                         "profiles/pmc_conv1ab.json instead and traffic_source says so")
    ap.add_argument("--no-measure-traffic", dest="measure_traffic", action="store_false")
    ap.add_argument("--library", default=None, help="developer A/B: load this build of the C-ABI library instead of the shipped one (explicit; the "
                                                    "package reads no environment variable for it).  The JSON line records it")
    ap.add_argument("--no-power", action="store_true", help="do not sample socket power / shader clock (scripts/power_telemetry.py)")
    ap.add_argument("--power-dump", default=None, help="write the poller's raw (t, W, MHz) samples + the stamps of the timed region to this JSON file")
    ap.add_argument("--stage-energy-s", type=float, default=0.6, help="seconds each isolated stage loops under the power poller (0 = skip the per-stage joules)")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: re-execute as N ranks under torch.distributed.run (does not return then)
    from superslam_amd.shard import relaunch_under_launcher_if_needed
    relaunch_under_launcher_if_needed(args.gpus, os.path.abspath(__file__), sys.argv[1:])

    import numpy as np
    import torch
    import torch.distributed as dist

    from superslam_amd import FrontEndBatch, LightGlue, SuperPoint, _lib
    from superslam_amd.synth import make_stereo_pair
    from superslam_amd.weights import make_lightglue_weights, make_superpoint_weights, save_safetensors

    _lib.set_library_path(args.library)   # None: the shipped superslam_amd/lib/libsuperslam_hip.so
    from superslam_amd.shard import all_reduce_max_seconds, dist_env, init_process_group

    rank, local_rank, world, under_launcher, backend = dist_env()   # local_rank = this rank's device (LOCAL_RANK unless pinned)
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks (WORLD_SIZE={world})")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library has no CPU path")
    torch.cuda.set_device(local_rank)
    os.environ.setdefault("SUPERSLAM_HIP_DEVICE", str(local_rank))
    use_dist = world > 1 or under_launcher   # under torch.distributed.run always go through RCCL, even at N = 1
    if use_dist:
        init_process_group(backend, local_rank)   # "nccl" = RCCL (SUPERSLAM_DIST_BACKEND=gloo: the one-GPU multi-rank rehearsal)
    _lib.init(local_rank)
    L = _lib.lib()

    import tempfile

    wdir = tempfile.mkdtemp(prefix=f"sship_bench_w{rank}_")
    spw, lgw = make_superpoint_weights(0), make_lightglue_weights(1)
    save_safetensors(spw, os.path.join(wdir, "sp.safetensors"))
    save_safetensors(lgw, os.path.join(wdir, "lg.safetensors"))
    P, CH = args.pairs, args.chunks
    sp = SuperPoint(os.path.join(wdir, "sp.safetensors"), args.max_kp, 0.005, 4, max_batch=2 * P)
    assert sp.initialize(), sp.last_error
    lg = LightGlue(os.path.join(wdir, "lg.safetensors"), W, H, max_keypoints=args.max_kp, max_pairs=P)
    assert lg.initialize(), lg.last_error

    # synthetic KITTI-shaped pairs (seed 1234 + pair index), uploaded once: inputs are HBM-resident when timing starts
    pairs = [make_stereo_pair(H, W, 1234 + 97 * rank + i) for i in range(P)]
    base = torch.from_numpy(np.stack([im for p in pairs for im in p])).cuda()
    chunks = make_chunks(torch, base, CH)
    fe = FrontEndBatch(sp, lg, P, H, W)
    # every library call of the bench goes on ONE stream: torch's current stream (the legacy default stream, handle 0)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        for x in chunks:
            fe.run(x, stream)

    # socket power / shader clock of rank 0's GPU, sampled by a thread while the steps run (reads of a sysfs file; the timed region itself
    # is untouched: same calls, same synchronisation)
    poller = None
    if rank == 0 and not args.no_power:
        sys.path.insert(0, os.path.join(ROOT, "scripts"))
        import power_telemetry

        poller = power_telemetry.PowerPoller(local_rank, hz=50.0)
        poller.__enter__()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tm0 = time.monotonic()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tm1 = time.monotonic()
    dt_own = dt
    if use_dist:
        dt = all_reduce_max_seconds(dt)   # the slowest rank defines the step time
    scale = scale_extras(torch, dist, rank, world, backend, dt_own, P * CH * args.steps) if (use_dist and world > 1) else None

    n_kp = fe.n.cpu().numpy()
    n_match = int((fe.matches0.cpu().numpy() >= 0).sum())
    total_pairs = world * P * CH * args.steps
    value = total_pairs / dt
    flops_pair = 2 * sp_flops_per_image(H, W) + lg_flops_per_pair(args.max_kp)

    if rank == 0:
        out = {
            "metric": "stereo pairs/sec (SPx2+LG) at 1376x376", "value": round(value, 2), "unit": "pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": f"configs[1]: SuperPoint x2 + select + gather + 1x LightGlue on 1376x376 stereo pairs; a step = "
                                   f"{CH} x {P} = {CH * P} pairs per GPU resident in HBM ({CH} library calls of {P} pairs, different images per "
                                   f"call), max_keypoints {args.max_kp}, seeded synthetic weights",
                       "pairs_per_step": P * CH, "pairs_per_call": P, "calls_per_step": CH, "max_keypoints": args.max_kp, "image": [H, W],
                       "keypoints_found": [int(n_kp.min()), int(n_kp.max())], "matches_last_call": n_match,
                       "timed_seconds": round(dt, 3),
                       "parallelism": f"replicated weights, pairs sharded over {world} rank(s), no data-path collective"
                                      + (f" (timing collectives on {backend})" if use_dist else "")},
            "algorithmic_gflop_per_pair": round(flops_pair / 1e9, 2),
            "effective_tflops": round(value * flops_pair / 1e12, 2),
            # the algorithmic count includes ~6 % FLOPs that are never executed (dense descriptor head replaced by the keypoint-only
            # head, out_proj / to_out folded into ffn.0): the executed figures are the ones to read against the MFMA peak
            "executed_gflop_per_pair": round(executed_flops_per_pair(H, W, args.max_kp) / 1e9, 2),
            "executed_tflops": round(value * executed_flops_per_pair(H, W, args.max_kp) / 1e12, 2),
            "executed_frac_of_mfma_peak": round(value * executed_flops_per_pair(H, W, args.max_kp) / 1e12 / MFMA_PEAK_TFLOPS, 4),
        }
        if args.library:
            out["library"] = "DEVELOPER BUILD " + os.path.relpath(_lib.LIB_PATH, ROOT) + " (not the shipped library: not a headline number)"
        if scale is not None:
            out["scale_extras"] = scale
        if poller is not None:
            import power_telemetry

            # this rank's GPU over the timed steps (the first 10 % of the window is dropped: the SMU's figure is a short moving average)
            pw = poller.window(tm0, tm1, settle_s=0.1 * (tm1 - tm0)) if poller.ok else None
            blk = power_telemetry.energy_block(pw, poller.cap_w, P * CH * args.steps / dt_own, "pair")
            if blk is not None:
                blk["backend"] = poller.backend.name
                blk["window"] = "the timed steps of rank 0's GPU (first 10 % dropped)"
                blk["value_over_pairs_per_s_at_cap"] = round((P * CH * args.steps / dt_own) / blk["pairs_per_s_at_cap"], 4) if blk.get("pairs_per_s_at_cap") else None
                out["power"] = blk
            else:
                out["power"] = {"error": "no power telemetry backend answered on this box (sysfs hwmon, librocm_smi64, rocm-smi)"}
        if not args.headline_only:
            out["self_check"] = self_check(torch, np, fe, chunks[CH - 1], stream, wdir, args.max_kp)
            extras(out, args, torch, np, L, sp, lg, fe, chunks, base, stream, wdir, world, spw, lgw, pairs, poller)
        if poller is not None:
            poller.__exit__(None, None, None)
            if args.power_dump and poller.ok:
                with open(args.power_dump, "w") as f:
                    json.dump({"backend": poller.backend.name, "cap_W": poller.cap_w, "timed_region": [tm0, tm1], "pairs_timed": P * CH * args.steps,
                               "samples": [[round(t, 4), w, c] for (t, w, c) in poller.samples]}, f)
        print(json.dumps(out), flush=True)
        if not args.headline_only and not out["self_check"]["ok"]:
            sp.close(); lg.close()
            raise SystemExit("bench.py self-check FAILED: the batched call and the per-pair path disagree: " + json.dumps(out["self_check"]))
    sp.close(); lg.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def extras(out, args, torch, np, L, sp, lg, fe, chunks, base, stream, wdir, world, spw, lgw, pairs, poller=None):
    """Everything next to the headline number (rank 0): per-stage times, rooflines, N = 1024, PCIe-inclusive variant, latency,
    the deployed unit, CPU baseline.  All after the timed region."""
    from superslam_amd import FrontEndBatch, LightGlue, SuperPoint, _lib

    P, CH, K = args.pairs, args.chunks, args.max_kp
    Hc, Wc = H // 8, W // 8
    B = 2 * P

    # developer aid (BENCH_LATENCY_PROBE=1): the P = 1 latency after every section of this function
    fe_probe = FrontEndBatch(sp, lg, 1, H, W) if os.environ.get("BENCH_LATENCY_PROBE") else None

    def lat_probe(tag):
        if fe_probe is None:
            return
        for _ in range(3):
            fe_probe.run(base[:2], stream)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            fe_probe.run(base[:2], stream)
        t_enq = time.perf_counter() - t
        torch.cuda.synchronize()
        out.setdefault("_latency_probe", {})[tag] = [round((time.perf_counter() - t) / 20 * 1e3, 3), round(t_enq / 20 * 1e3, 3)]

    lat_probe("start")
    # ---- step-time distribution: 40 calls, events between calls on the launch stream, no host synchronisation ----
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(41)]
    ev[0].record()
    for i in range(40):
        fe.run(chunks[i % CH], stream)
        ev[i + 1].record()
    torch.cuda.synchronize()
    call_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(40))
    out["call_ms"] = {"calls": 40, "pairs_per_call": P, "median": round(call_ms[20], 4), "p95": round(call_ms[37], 4),
                      "min": round(call_ms[0], 4), "max": round(call_ms[-1], 4),
                      "note": "BASELINE.md / SURVEY 8(d) config 2 asks for 50 warm-up + 500 timed PAIRS with median and p95; the unit timed here is a "
                              f"library CALL of {P} pairs (device events between calls, no host synchronisation): {40 * P} timed pairs after "
                              f"{args.warmup * CH * P + args.steps * CH * P} pairs of warm-up + headline steps; per-pair figures are these divided by {P}; "
                              "the single-pair unit itself is `latency_ms_single_pair` (median of five 50-call blocks)"}

    lat_probe("after_call_ms")
    # ---- per-stage device time (hipEvents inside the library): IN-SITU launch durations of profiled headline calls ----
    # level 2 = one event per SuperPoint layer launch ("<scope>:<stage>/<layer>"); mean over `reps` calls on different chunks.
    # These are the durations the kernels have INSIDE the pipeline (what rocprofv3 --kernel-trace of the headline steps shows,
    # profiles/r03_*_kernel_stats_P64.txt) - the roofline line below is priced on them.
    L.sship_set_profiling(2)
    reps, acc = 6, {}
    for i in range(reps):
        fe.run(chunks[i % CH], stream); torch.cuda.synchronize()
        for k, v in _lib.stage_timings().items():
            acc[k] = acc.get(k, 0.0) + v / reps
    L.sship_set_profiling(0)
    insitu = {k.split("/", 1)[1]: round(v, 4) for k, v in acc.items() if "/" in k}
    stages = {}
    for k, v in acc.items():    # level-1 stage = sum of its launches
        stages[k.split("/")[0]] = round(stages.get(k.split("/")[0], 0.0) + v, 4)
    scopes = {}
    for k, v in stages.items():   # the reference's own SUPERSLAM_PROFILE labels = sums over this library's finer stages
        scopes[k.split(":")[0]] = round(scopes.get(k.split(":")[0], 0.0) + v, 4)
    scopes["fe_extract_stereo"] = round(scopes.get("sp_gpu_infer", 0.0) + scopes.get("sp_extract_stereo", 0.0), 4)
    out["stage_ms"] = stages
    out["insitu_launch_ms"] = insitu
    out["reference_scope_ms"] = scopes

    lat_probe("after_insitu")
    # ---- matrix-core stages: launch time by HIP events on the kernel's own stream -> TFLOP/s vs the dense fp16 peak ----
    def sp_layer(lid, iters=10):
        ms, macs = C.c_float(0), C.c_double(0)
        _lib.check(L.sship_sp_bench_layer(sp._h, lid, B, H, W, iters, C.byref(ms), C.byref(macs)))
        return ms.value, macs.value

    def lg_stage(sid, iters=10):
        ms = C.c_float(0)
        _lib.check(L.sship_lg_bench_stage(lg._h, sid, iters, C.byref(ms)))
        return ms.value

    # roofline of the dominant kernel.  `frac` is priced on the IN-SITU launch (events around the launch inside profiled headline
    # calls, above); `frac_isolated` on the same launch repeated 20x back to back on the same real pixels (sship_sp_bench_layer;
    # a profiled call keeps a copy of its input - round 2 re-launched on a zero image, which clocks ~7 % higher: the chip runs at
    # its power limit and low-toggle operands draw less, MI355X_MICROARCH.md "DVFS give-back").
    L.sship_set_profiling(1)                         # the handle keeps a copy of this call's pixels and activations: what the
    fe.run(chunks[0], stream); torch.cuda.synchronize()   # isolated re-launches below run on (sship_sp_bench_layer refuses stale pixels)
    _lib.stage_timings()
    L.sship_set_profiling(0)
    ms1_iso, macs1 = sp_layer(1, 20)
    ms1 = insitu.get("conv1a+conv1b+pool", ms1_iso)
    ach = 2.0 * macs1 / (ms1 * 1e-3) / 1e12
    ach_iso = 2.0 * macs1 / (ms1_iso * 1e-3) / 1e12
    traffic, traffic_source = None, None
    pmc_kernels = {}
    pmc = os.path.join(ROOT, "profiles", "pmc_conv1ab.json")
    if os.path.exists(pmc):
        try:
            import hashlib

            pj = json.load(open(pmc))
            if pj.get("pairs_per_call", pj.get("pairs_per_step")) == P and pj.get("headline_launches_only"):
                traffic = pj.get("hbm_bytes_per_launch")
                cur = hashlib.sha256(open(os.path.join(ROOT, "superslam_amd", "csrc", "conv_pp.hip"), "rb").read()).hexdigest()[:16]
                # a FLAT string: the driver's record flattens `roofline` and dropped the nested dict round 3 used
                traffic_source = ("READ from profiles/pmc_conv1ab.json, NOT measured in this run; collected " + str(pj.get("collected_at"))
                                  + " by scripts/pmc_traffic.sh (two separate rocprofv3 --pmc passes over `bench.py --headline-only`, "
                                  "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 per MI355X_MICROARCH.md); conv_pp.hip sha16 then "
                                  + str(pj.get("conv_pp_sha16")) + " now " + cur
                                  + (" (kernel source unchanged)" if pj.get("conv_pp_sha16") == cur else " (KERNEL SOURCE CHANGED since the collection)"))
        except Exception:
            traffic, traffic_source = None, None
    import shutil

    under_profiler = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", "")  # never nest profilers
    if getattr(args, "measure_traffic", False) and world == 1 and shutil.which("rocprofv3") and not under_profiler:
        # the PMC passes run NOW, as child processes on the same GPU, and the figure below is this run's
        import subprocess

        mj = os.path.join(wdir, "pmc_traffic_now.json")
        try:
            import signal

            pr = subprocess.Popen(["bash", os.path.join(ROOT, "scripts", "pmc_traffic.sh"), str(P), mj], cwd=ROOT, stdout=subprocess.DEVNULL,
                                  stderr=subprocess.DEVNULL, start_new_session=True)   # its own process group: a timeout takes rocprofv3 and its python along
            try:
                pr.wait(timeout=150)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)
                pr.wait()
                raise
            if pr.returncode != 0:
                raise RuntimeError(f"scripts/pmc_traffic.sh exited with {pr.returncode}")
            pmc_kernels.update(json.load(open(mj)).get("kernels", {}))     # every kernel of the headline steps: the memory-bound rows below use theirs
            kj = json.load(open(os.path.join(ROOT, "gpurun_out", "pmc_conv1ab.json")))
            if kj.get("pairs_per_call") == P and kj.get("headline_launches_only"):
                traffic = kj.get("hbm_bytes_per_launch")
                traffic_source = ("MEASURED IN THIS RUN: scripts/pmc_traffic.sh as a child process (two separate rocprofv3 --pmc passes over `bench.py --headline-only`, "
                                  "(2 x FETCH_SIZE + WRITE_SIZE) x 1024 per launch per MI355X_MICROARCH.md), collected " + str(kj.get("collected_at")))
        except Exception as e:  # noqa: BLE001 - fall back to the file figure already in `traffic`
            traffic_source = (traffic_source or "") + f" [--measure-traffic failed: {type(e).__name__}: {e}]"[:200]
    alg_bytes = B * (H * W + (H // 2) * (W // 2) * 64 * 2)   # u8 image in, pooled fp16 64-ch map out
    probe = {}
    for name, rnd in (("zero_operands", 0), ("random_operands", 1)):
        tf = C.c_float(0)
        _lib.check(L.sship_mfma_probe(rnd, C.byref(tf)))
        probe[name] = round(tf.value, 1)
    out["roofline"] = {"kernel": "conv3x3_pp<64,64,pool,fuse1a> (conv1a+conv1b+maxpool, 36 % of the pair's FLOPs)", "bound": "mfma",
                       "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                       "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_source,
                       "launch_ms": round(ms1, 4), "launch_ms_source": "in-situ: hipEvents around the launch inside profiled headline calls (mean of 6)",
                       "launch_ms_isolated": round(ms1_iso, 4), "achieved_isolated": round(ach_iso, 2),
                       "frac_isolated": round(ach_iso / MFMA_PEAK_TFLOPS, 4),
                       "flops_per_launch": 2.0 * macs1, "images_per_launch": B,
                       "algorithmic_bytes_per_launch": alg_bytes,
                       "sustained_mfma_probe_tflops": probe,
                       "frac_of_sustained_random_probe": round(ach / probe["random_operands"], 4) if probe["random_operands"] > 0 else None}
    # Every conv row also carries the HBM side of its roofline: algorithmic bytes per launch (activation in + activation out, fp16
    # channels-last; weights are KBs), achieved GB/s and the bound chosen by FLOP/B against the ridge 2500 TFLOP/s / 8 TB/s = 312.5 FLOP/B.
    # conv2a (288 FLOP/B) and conv3a / conv4a / conv4b sit AT the ridge: their 0.45 of the MFMA peak is also ~0.5 of the HBM peak
    #  - `frac` alone reads them as weak matrix kernels, they are nearly saturated memory pipes.
    RIDGE = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    H2, W2, H4, W4 = H // 2, W // 2, H // 4, W // 4
    conv_bytes = {"conv1a+conv1b+pool": B * (H * W + H2 * W2 * 64 * 2), "conv2a": B * H2 * W2 * 64 * 2 * 2,
                  "conv2b+pool": B * (H2 * W2 + H4 * W4) * 64 * 2, "conv3a": B * H4 * W4 * (64 + 128) * 2,
                  "conv3b+pool": B * (H4 * W4 + Hc * Wc) * 128 * 2, "conv4a": B * Hc * Wc * 128 * 2 * 2, "conv4b": B * Hc * Wc * 128 * 2 * 2,
                  "convPa": B * Hc * Wc * (128 + 256) * 2}

    def classify(entry, flops, nbytes, ms):
        inten = flops / nbytes
        gbs = nbytes / (ms * 1e-3) / 1e9
        entry.update({"algorithmic_bytes_per_launch": int(nbytes), "flop_per_byte": round(inten, 1), "ridge_flop_per_byte": round(RIDGE, 1),
                      "achieved_gb_per_s": round(gbs, 1), "frac_of_hbm_peak": round(gbs / HBM_PEAK_GBS, 4),
                      "bound": "mfma" if inten >= 2 * RIDGE else ("hbm" if inten <= RIDGE / 2 else "hbm+mfma (ridge)")})

    mfma = []
    # Throughput batches run conv2a -> conv2b -> pool as ONE launch (csrc/conv_fuse2.hip, layer id 15): the profiled calls above then carry a
    # "conv2a+conv2b+pool" mark and the two separate launches (ids 2, 3: what a one-pair call runs) are off the path of this workload.
    fused2 = "conv2a+conv2b+pool" in insitu
    names = ["conv1a_standalone_offpath", "conv1a+conv1b+pool", "conv2a" + ("_split_offpath" if fused2 else ""),
             "conv2b+pool" + ("_split_offpath" if fused2 else ""), "conv3a", "conv3b+pool", "conv4a", "conv4b",
             "convPa", "convPb", "convDa_dense_offpath", "convDb_dense_offpath"]
    sp_layer_ids = {name: lid for lid, name in enumerate(names)}
    if fused2:
        names.insert(2, "conv2a+conv2b+pool")
        sp_layer_ids["conv2a+conv2b+pool"] = 15
        conv_bytes["conv2a+conv2b+pool"] = B * (H2 * W2 + H4 * W4) * 64 * 2    # the map between the two layers never reaches HBM
    layer_ms = {}
    for name in names:
        lid = sp_layer_ids[name]
        ms, macs = sp_layer(lid)
        layer_ms[name] = round(ms, 4)
        if "offpath" not in name and lid != 9:
            ms_in = insitu.get(name, ms)    # in-situ launch duration where the profiled calls recorded one
            tf = 2.0 * macs / (ms_in * 1e-3) / 1e12
            mfma.append({"kernel": name, "launch_ms": round(ms_in, 4), "launch_ms_isolated": round(ms, 4),
                         "gflop_per_launch": round(2.0 * macs / 1e9, 2),
                         "achieved": round(tf, 1), "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
                         "frac_isolated": round(2.0 * macs / (ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)})
            if name in conv_bytes:
                classify(mfma[-1], 2.0 * macs, conv_bytes[name], ms_in)
    out["layer_ms"] = layer_ms
    # LightGlue stages over the state of the last call (P pairs, S = 2P sequences of n keypoints each; FLOPs for n = max_kp)
    n, S = K, 2 * P
    lg_flops = {
        "lg_wqkv0_proj": 2.0 * S * n * 768 * 256,
        "lg_self_attention": 2.0 * S * 4 * n * n * 64 * 2,
        "lg_cross_attention": 2.0 * S * 4 * n * n * 64 * 2,
        "lg_self_ffn+to_qk|to_v": 2.0 * S * n * (512 * 512 + 512 * 256 + 256 * 256 + 512 * 256),
        "lg_cross_ffn+wqkv": 2.0 * S * n * (512 * 512 + 512 * 256 + 256 * 256 + 768 * 256),
        "lg_last_ffn+final_proj": 2.0 * S * n * (512 * 512 + 512 * 256 + 256 * 256 + 256 * 256 + 256),
        "lg_assign_pass1_lse": 2.0 * P * n * n * 256,   # sim tiles (algorithmic count: once) + row / column log-sum-exp
        "lg_assign_pass2_argmax": 2.0 * P * n * n * 256,  # sim tiles again + row / column arg-max
    }
    lg_ms = {}
    for sid, name in enumerate(lg_flops):
        ms = lg_stage(sid)
        lg_ms[name] = round(ms, 4)
        tf = lg_flops[name] / (ms * 1e-3) / 1e12
        mfma.append({"kernel": name, "launch_ms": round(ms, 4), "gflop_per_launch": round(lg_flops[name] / 1e9, 2),
                     "achieved": round(tf, 1), "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
                     "timing": "isolated re-launch over the state of the last call (the call runs two half-batches on two streams: no "
                               "in-situ per-launch figure exists)",
                     "note": "out_proj / to_out are folded into ffn.0 on the host: their FLOPs are counted (algorithmic), not executed"
                     if "ffn" in name else None})
    out["roofline_mfma"] = mfma
    lg_layers_ms = stages.get("fe_lg_stereo_match:layers_x9", 0.0)
    lg_total_ms = sum(v for k, v in stages.items() if k.startswith("fe_lg_stereo_match"))
    if lg_total_ms > 0:
        out["lightglue_mfma"] = {"gflop_per_call": round(P * lg_flops_per_pair(K) / 1e9, 1), "ms_per_call": round(lg_total_ms, 4),
                                 "layers_x9_ms": round(lg_layers_ms, 4),
                                 "achieved": round(P * lg_flops_per_pair(K) / (lg_total_ms * 1e-3) / 1e12, 1), "unit": "TFLOP/s",
                                 "frac": round(P * lg_flops_per_pair(K) / (lg_total_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4)}

    lat_probe("after_mfma_stages")
    # ---- memory-bound stages: algorithmic bytes / launch time vs 8 TB/s ----
    n_cand_bytes = 8.0 * 7000  # ~7 k candidates x 8 B per image (data dependent)
    hbm = []

    def hbm_entry(kernel, ms, alg_bytes, what, impl_bytes=None):
        gbs = alg_bytes / (ms * 1e-3) / 1e9
        e = {"kernel": kernel, "bound": "hbm", "launch_ms": round(ms, 4), "algorithmic_bytes_per_launch": int(alg_bytes),
             "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4), "bytes": what}
        if impl_bytes is not None:
            e["implementation_bytes_per_launch"] = int(impl_bytes)
        hbm.append(e)

    ms_nms, _ = sp_layer(12)
    hbm_entry("k_nms_tile (softmax + depth-to-space + 9x9 NMS + threshold + compaction)", ms_nms,
              B * (65 * Hc * Wc * 4 + n_cand_bytes), "fp32 logits [65,Hc,Wc] read once + ~7k candidates x 8 B written, per image",
              B * (68 * Hc * Wc * 4 + n_cand_bytes))
    ms_pb = layer_ms["convPb"]
    hbm_entry("k_convpb_stream convPb (1x1, 256 -> 65, fp32 logits)", ms_pb, B * (256 * Hc * Wc * 2 + 65 * Hc * Wc * 4),
              "fp16 convPa map read + fp32 logits written, per image", B * (256 * Hc * Wc * 2 + 80 * Hc * Wc * 4))
    ms_topk, _ = sp_layer(13)
    hbm_entry("k_topk (radix select + bitonic sort + keypoints/cells)", ms_topk, B * (n_cand_bytes + K * 20),
              "candidates read once + kp/cells written (latency-bound: one workgroup per image)")
    ms_dh, _ = sp_layer(14)
    hbm_entry("k_desc_head_sparse (convDa + convDb at the keypoints + normalise x2 + gather)", ms_dh,
              B * (K * 9 * 128 * 2 + K * 256 * 2 + 8 * K), "9 x 256 B encoder rows per keypoint read + [N,256] fp16 written + 8N index bytes, "
              "per image (SURVEY 8(d) gather figure is the last two terms: N*256*2 read + write + 8N)")
    ms_as = lg_stage(6) + lg_stage(7)
    hbm_entry("k_assign_stream x2 + mutual filter (log-sum-exp pass, arg-max pass; no sim matrix in memory)", ms_as,
              P * (2 * 2 * n * 256 * 2 + 4 * n * 4),
              "final projections of both images [n,256] fp16 read once per pass + match vectors, per pair (matrix-pipe / VALU bound)",
              P * 2 * ((n + 31) // 32 + 4) * n * 256 * 2)
    # the assignment recomputes its similarity tiles on the matrix pipe instead of streaming a stored matrix: its bound is the
    # matrix pipe + VALU (it is priced in roofline_mfma as lg_assign_pass1 / pass2); the entry above is kept for its byte counts
    hbm[-1]["bound"] = "mfma+valu (by design: 11x the algorithmic bytes are re-read from L2 instead of a 92 MB fp32 matrix going through HBM four times)"
    # counter-derived HBM bytes per launch next to the stated implementation bytes.  FETCH_SIZE x 2 holds for these kernels'
    # access patterns too: scripts/ubench/fetch_calib.hip reads a known byte count exactly once in k_nms_tile's own per-lane pattern and the
    # counter reports 0.5001 of it (profiles/r06_e_fetch_size_calibration.json).  Round 5's 590 MB per launch of k_nms_tile (2.1 x) was REAL:
    # every 4 x 8-cell tile re-reads its one-cell halo ring (60 cells for 32) and neighbouring tiles ran on different XCDs.  With the XCD-aware
    # tile order of round 6 the ring is an L2 hit: 332 MB = 1.15 x the implementation bytes, 258 -> 153 us (profiles/r06_f_nms_xcd_tile_order.txt).
    for row, key in ((hbm[0], "k_nms_tile"), (hbm[1], "k_convpb_stream"), (hbm[2], "k_topk"), (hbm[3], "k_desc_head_sparse")):
        hit = [v for k, v in pmc_kernels.items() if key in k]
        if hit:
            row["traffic"] = hit[0]["hbm_bytes_per_launch"]
            row["traffic_source"] = "measured in this run: (2 x FETCH_SIZE + WRITE_SIZE) x 1024 of the headline steps' launches (scripts/pmc_traffic.sh)"
            row["traffic_over_algorithmic"] = round(hit[0]["hbm_bytes_per_launch"] / row["algorithmic_bytes_per_launch"], 3)
            if "implementation_bytes_per_launch" in row:
                row["traffic_over_implementation"] = round(hit[0]["hbm_bytes_per_launch"] / row["implementation_bytes_per_launch"], 3)
    hbm[0]["traffic_note"] = ("a 4 x 8-cell tile loads its one-cell halo ring (6 x 10 cells = 1.83 x the cells); the XCD-aware tile order makes the ring an L2 "
                              "hit (round 5: 590 MB per launch at the fabric); FETCH_SIZE x 2 calibrated on this access pattern (profiles/r06_e_fetch_size_calibration.json: 0.5001)")
    out["roofline_hbm"] = hbm

    # ---- joules per launch: every stage loops for >= --stage-energy-s seconds under the poller; energy = mean socket power x launch time ----
    # (gross socket power, idle floor included - the same convention as `power.joules_per_pair`, so the rows add up to the call)
    if poller is not None and poller.ok and args.stage_energy_s > 0:
        def stage_energy(run_iters, ms_guess):
            iters = max(10, int(args.stage_energy_s * 1e3 / max(ms_guess, 1e-3)))
            ta = time.monotonic()
            ms = run_iters(iters)
            tb = time.monotonic()
            w = poller.window(ta, tb, settle_s=0.25 * (tb - ta))
            if not w:
                return None
            e = {"joules_per_launch": round(w["avg_W"] * ms * 1e-3, 5), "avg_W": w["avg_W"], "sclk_MHz": w["sclk_MHz"],
                 "launch_ms_energy_loop": round(ms, 4), "samples": w["n"]}
            # the MEASURED bound next to the FLOP/B classification (`bound`): a stage that loops at >= 93 % of the board's cap (the SMU holds the
            # average a few per cent under the limit while it throttles the clock: conv1a+1b reads 1 330-1 385 W at 1.9 of 2.4 GHz) is power-bound -
            # its launch time is its joules divided by the cap, whatever the two datasheet roofs say
            if poller.cap_w:
                e["bound_measured"] = ("power cap (time = joules / cap)" if w["avg_W"] >= 0.93 * poller.cap_w
                                       else f"below the cap ({w['avg_W'] / poller.cap_w:.2f} of it): latency / memory / issue bound")
            return e

        sp_ids = sp_layer_ids
        lg_ids = {name: sid for sid, name in enumerate(lg_flops)}
        for row in out["roofline_mfma"]:
            k = row["kernel"]
            e = None
            if k in sp_ids:
                e = stage_energy(lambda it, lid=sp_ids[k]: sp_layer(lid, it)[0], row.get("launch_ms_isolated", row["launch_ms"]))
            elif k in lg_ids:
                e = stage_energy(lambda it, sid=lg_ids[k]: lg_stage(sid, it), row["launch_ms"])
            if e:
                row.update(e)
        for row, lid in ((hbm[0], 12), (hbm[2], 13), (hbm[3], 14)):
            e = stage_energy(lambda it, lid=lid: sp_layer(lid, it)[0], row["launch_ms"])
            if e:
                row.update(e)
        e = stage_energy(lambda it: sp_layer(9, it)[0], hbm[1]["launch_ms"])
        if e:
            hbm[1].update(e)
        # the call's joule budget from its launches (one 64-pair call = the SuperPoint launches once + the LightGlue stage mix), next to the
        # figure measured over the timed steps: they agree when nothing but these launches draws power
        per = {r["kernel"]: r.get("joules_per_launch") for r in out["roofline_mfma"] + hbm}
        if all(per.get(k) is not None for k in ("conv1a+conv1b+pool", "lg_self_attention", "lg_self_ffn+to_qk|to_v")):
            sp_j = sum(per[k] for k in ("conv1a+conv1b+pool", "conv2a", "conv2b+pool", "conv2a+conv2b+pool", "conv3a", "conv3b+pool", "conv4a", "conv4b", "convPa") if per.get(k))
            sp_j += sum(v for k, v in per.items() if v and k.startswith(("k_nms_tile", "k_convpb_stream", "k_topk", "k_desc_head_sparse")))
            lg_j = (per["lg_wqkv0_proj"] + 9 * per["lg_self_attention"] + 9 * per["lg_cross_attention"] + 9 * per["lg_self_ffn+to_qk|to_v"]
                    + 8 * per["lg_cross_ffn+wqkv"] + per["lg_last_ffn+final_proj"] + per["lg_assign_pass1_lse"] + per["lg_assign_pass2_argmax"])
            out["joule_budget_per_call"] = {"superpoint_J": round(sp_j, 3), "lightglue_J": round(lg_j, 3), "sum_J": round(sp_j + lg_j, 3),
                                            "per_pair_J": round((sp_j + lg_j) / P, 5),
                                            "measured_over_timed_steps_per_pair_J": out.get("power", {}).get("joules_per_pair"),
                                            "note": "sum over the call's launches of (mean socket power of that stage looping alone) x (its launch time); the two "
                                                    "half-batch streams of the LightGlue call overlap in time, not in joules"}

    lat_probe("after_hbm_stages")
    # ---- N = 1024 keypoints per image (the reference engine's upper profile; SURVEY 8(d) config 2 second run) ----
    K2 = 1024
    sp2 = SuperPoint(os.path.join(wdir, "sp.safetensors"), K2, 0.005, 4, max_batch=B)
    lg2 = LightGlue(os.path.join(wdir, "lg.safetensors"), W, H, max_keypoints=K2, max_pairs=P)
    assert sp2.initialize() and lg2.initialize()
    fe2 = FrontEndBatch(sp2, lg2, P, H, W)
    for x in chunks[:2]:
        fe2.run(x, stream)
    torch.cuda.synchronize()
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        for x in chunks:
            fe2.run(x, stream)
    torch.cuda.synchronize()
    dt2 = time.perf_counter() - t0
    n2 = fe2.n.cpu().numpy()
    fp2 = 2 * sp_flops_per_image(H, W) + lg_flops_per_pair(K2)
    out["n1024"] = {"value": round(reps * CH * P / dt2, 2), "unit": "pairs/s", "max_keypoints": K2,
                    "keypoints_found": [int(n2.min()), int(n2.max())], "pairs_timed": reps * CH * P, "timed_seconds": round(dt2, 3),
                    "algorithmic_gflop_per_pair": round(fp2 / 1e9, 2), "effective_tflops": round(reps * CH * P / dt2 * fp2 / 1e12, 2),
                    "matches_last_call": int((fe2.matches0 >= 0).sum().item())}
    # the matcher's stages at 1 024 keypoints (the quadratic attention term is 2.9x the 600-keypoint one): isolated re-launches over lg2's last call
    n2k, rows1024 = K2, []
    fl2 = {"lg_self_attention": 2.0 * S * 4 * n2k * n2k * 64 * 2, "lg_cross_attention": 2.0 * S * 4 * n2k * n2k * 64 * 2,
           "lg_self_ffn+to_qk|to_v": 2.0 * S * n2k * (512 * 512 + 512 * 256 + 256 * 256 + 512 * 256),
           "lg_cross_ffn+wqkv": 2.0 * S * n2k * (512 * 512 + 512 * 256 + 256 * 256 + 768 * 256)}
    for sid, name in ((1, "lg_self_attention"), (2, "lg_cross_attention"), (3, "lg_self_ffn+to_qk|to_v"), (4, "lg_cross_ffn+wqkv")):
        msv = C.c_float(0)
        _lib.check(L.sship_lg_bench_stage(lg2._h, sid, 10, C.byref(msv)))
        tf = fl2[name] / (msv.value * 1e-3) / 1e12
        rows1024.append({"kernel": name + " @ n = 1024", "launch_ms": round(msv.value, 4), "gflop_per_launch": round(fl2[name] / 1e9, 2),
                         "achieved": round(tf, 1), "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
                         "timing": "isolated re-launch over the state of the last 1 024-keypoint call"})
    out["roofline_mfma"].extend(rows1024)
    sp2.close(); lg2.close()

    lat_probe("after_n1024")
    # ---- PCIe-inclusive variant: u8 images from pinned host memory (double buffered on a copy stream), keypoints / counts /
    # matches / scores copied back to pinned memory, all inside the timed region.  Descriptors stay on the device, as in
    # the reference (pool slots, DescriptorPool.h). ----
    host_in = [x.cpu().pin_memory() for x in chunks]
    dev_in = [torch.empty_like(chunks[0]) for _ in range(2)]
    h_kp = torch.empty((B, K, 3), dtype=torch.float32).pin_memory()
    h_n = torch.empty((B,), dtype=torch.int32).pin_memory()
    h_m = torch.empty((P, K), dtype=torch.int32).pin_memory()
    h_s = torch.empty((P, K), dtype=torch.float32).pin_memory()
    copy_s = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()
    up = [torch.cuda.Event() for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]

    def e2e_pass(reps):
        with torch.cuda.stream(copy_s):
            dev_in[0].copy_(host_in[0], non_blocking=True); up[0].record(copy_s)
        for i in range(reps * CH):
            b = i & 1
            if i + 1 < reps * CH:
                with torch.cuda.stream(copy_s):
                    copy_s.wait_event(done[b ^ 1]) if i >= 1 else None   # the buffer's previous consumer has finished
                    dev_in[b ^ 1].copy_(host_in[(i + 1) % CH], non_blocking=True); up[b ^ 1].record(copy_s)
            main_s.wait_event(up[b])
            fe.run(dev_in[b], stream)
            done[b].record(main_s)
            h_kp.copy_(fe.kp, non_blocking=True); h_n.copy_(fe.n, non_blocking=True)
            h_m.copy_(fe.matches0, non_blocking=True); h_s.copy_(fe.mscores0, non_blocking=True)
        torch.cuda.synchronize()

    e2e_pass(1)
    reps = 4
    t0 = time.perf_counter()
    e2e_pass(reps)
    dte = time.perf_counter() - t0
    h2d = reps * CH * B * H * W
    d2h = reps * CH * (B * K * 12 + B * 4 + P * K * 8)
    out["end_to_end"] = {"value": round(reps * CH * P / dte, 2), "unit": "pairs/s", "pairs_timed": reps * CH * P, "timed_seconds": round(dte, 3),
                         "includes": "pinned u8 H2D (double buffered, copy stream) + fused step + D2H of keypoints, counts, matches0, mscores0",
                         "h2d_bytes_per_pair": 2 * H * W, "d2h_bytes_per_pair": int(d2h / (reps * CH * P)),
                         "h2d_gb_per_s": round(h2d / dte / 1e9, 2), "matches_last_call": int((h_m >= 0).sum().item())}

    lat_probe("after_e2e")
    # ---- single-pair latency (the reference's per-frame unit): P = 1 through the same fused call ----
    fe1 = FrontEndBatch(sp, lg, 1, H, W)
    for _ in range(5):
        fe1.run(base[:2], stream)
    torch.cuda.synchronize()
    blocks = []
    for _ in range(5):     # median of five 50-call blocks (a single block once read 2.0 ms on an otherwise 0.8 ms box: a host hiccup)
        t1 = time.perf_counter()
        for _ in range(50):
            fe1.run(base[:2], stream)
        torch.cuda.synchronize()
        blocks.append((time.perf_counter() - t1) / 50 * 1e3)
    blocks.sort()
    out["latency_ms_single_pair"] = round(blocks[2], 4)
    out["latency_ms_single_pair_blocks"] = [round(b, 4) for b in blocks]
    # BASELINE.md / SURVEY 8(d) config 2 to the letter, on the per-frame unit: 50 warm-up pairs + 500 timed pairs, one pair per call,
    # device events around every call (no host synchronisation inside the series), median and p95 per pair
    for _ in range(50):
        fe1.run(base[:2], stream)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(501)]
    evs[0].record()
    for i in range(500):
        fe1.run(base[:2], stream)
        evs[i + 1].record()
    torch.cuda.synchronize()
    per = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(500))
    out["single_pair_protocol"] = {"warmup_pairs": 50, "timed_pairs": 500, "median_ms": round(per[250], 4), "p95_ms": round(per[474], 4),
                                   "pairs_per_s_at_median": round(1e3 / per[250], 1),
                                   "unit": "one 1376x376 stereo pair per library call (SuperPoint batch 2 + select + descriptor head + 1x LightGlue), "
                                           "device-resident, back to back on one stream"}

    # ---- the unit the reference actually runs per frame: the stereo match PLUS a second LightGlue call
    # against the previous keyframe (VoEstimator.cc:243).  Emulated with left(p) vs left(p+1) on the features of the call
    # itself; both LightGlue calls and the extractor are on the same stream.  Reported next to the headline, not instead. ----
    idx = torch.arange(2 * P, device="cuda").view(P, 2)
    idx[:, 1] = (idx[:, 0] + 2) % (2 * P)      # set 0 = left of pair p, set 1 = left of pair p + 1
    idx = idx.reshape(-1)
    m2 = torch.empty((P, K), dtype=torch.int32, device="cuda")
    s2 = torch.empty((P, K), dtype=torch.float32, device="cuda")

    def deployed_call(x):
        fe.run(x, stream)
        lg.match_batch_device(fe.kp.index_select(0, idx), fe.n.index_select(0, idx), fe.desc.index_select(0, idx), m2, s2, stream)

    for x in chunks[:2]:
        deployed_call(x)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(3):
        for x in chunks:
            deployed_call(x)
    torch.cuda.synchronize()
    out["deployed_unit"] = {"pairs_per_s": round(3 * CH * P / (time.perf_counter() - t2), 2),
                            "unit": "SuperPoint x2 + LightGlue(L,R) + LightGlue(L, previous keyframe L) per pair",
                            "keyframe_matches_last_call": int((m2 >= 0).sum().item())}
    # ---- EigenPlaces global descriptor (once per keyframe on the loop-closure thread, not part of `value`) ----
    from superslam_amd import EigenPlaces
    from superslam_amd.weights import make_eigenplaces_weights

    epp = os.path.join(wdir, "eigenplaces.safetensors")
    save_ep = __import__("superslam_amd.weights", fromlist=["save_safetensors"]).save_safetensors
    save_ep(make_eigenplaces_weights(2), epp)
    ep = EigenPlaces(epp, 512, 512)
    assert ep.initialize(), ep.last_error
    ep.compute_global_descriptor(pairs[0][0])
    ep_blocks = []
    for _ in range(5):      # median of five 10-call blocks: a single 20-call block once read 0.83 ms on a 0.29-ms box (a host hiccup, as with the pair latency)
        t3 = time.perf_counter()
        for _ in range(10):
            ep.compute_global_descriptor(pairs[0][0])
        ep_blocks.append((time.perf_counter() - t3) / 10 * 1e3)
    ep_ms = sorted(ep_blocks)[2]
    ep_dev = C.c_float(0)
    dimg = torch.from_numpy(pairs[0][0]).cuda()
    _lib.check(L.sship_ep_bench(ep._h, dimg.data_ptr(), H, W, W, 1, 20, C.byref(ep_dev)))
    out["eigenplaces"] = {"ms_per_descriptor": round(ep_ms, 3),
                          "includes": "u8 H2D (0.5 MB, pinned) + device resize to 512x512 / normalise + ResNet-18 + GeM + FC + D2H, synchronous (sship_ep_infer_u8)",
                          "ms_per_descriptor_device_resident": round(ep_dev.value, 4), "gflop": 19.0, "input": [H, W],
                          "achieved_tflops_device_resident": round(19.0 / ep_dev.value, 1),
                          "frac_of_mfma_peak_device_resident": round(19.0 / ep_dev.value / MFMA_PEAK_TFLOPS, 4),
                          "ms_per_descriptor_blocks": [round(b, 3) for b in sorted(ep_blocks)],
                          "bound": "launch latency: ~36 dependent launches of 3-18 us on ONE 512x512 image (per-kernel table: profiles/r06_final_ep_kernel_stats.txt)"}
    out["roofline_mfma"].append({"kernel": "eigenplaces (ResNet-18 + GeM + FC, one 512x512 image, whole descriptor)", "launch_ms": round(ep_dev.value, 4),
                                 "gflop_per_launch": 19.0, "achieved": round(19.0 / ep_dev.value, 1), "unit": "TFLOP/s",
                                 "frac": round(19.0 / ep_dev.value / MFMA_PEAK_TFLOPS, 4),
                                 "timing": "sship_ep_bench: 20 back-to-back device-resident descriptors on the handle's stream (not one launch: ~45 kernels)",
                                 "bound": "launch latency at batch 1 (off the per-frame path: once per keyframe on the loop-closure thread)"})
    ep.close()
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(spw, lgw, pairs[0][0], pairs[0][1], K)
    # Real checkpoints (SUPERSLAM_SP_WEIGHTS / SUPERSLAM_LG_WEIGHTS set): the dry-run kit's verdict next to the numbers, which are ALL on
    # seeded weights (a checker leg like cpu_baseline: a child process, after the timed region; absent when the variables are unset)
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import real_weights_check

    rw = real_weights_check.verdict_from_env()
    if rw is not None:
        out["real_weights"] = rw


if __name__ == "__main__":
    main()
