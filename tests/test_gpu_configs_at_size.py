"""GPU (-m gpu): BASELINE configs[2] and configs[4] AT THEIR STATED SIZES, against their stated pass criteria (BASELINE.md 4 rows 3 and 5;
VERDICT r05 "What's missing" 4).  One GPU here, so the N-rank form is rehearsed with two ranks on device 0 (gloo - RCCL refuses two ranks on
one device; tests/test_gpu_multirank_rehearsal.py); the single-rank form runs the same scripts at world size 1.

  configs[4]  eight 1280 x 720 streams, max_kp 1024, 28 cross-camera LightGlue pairs over the all-gathered descriptors:
              "matches equal single-GPU run" - every pair's matches0 / mscores0 out of scripts/multicam.py equals the same pair matched
              directly from per-camera `extract` results: bit for bit when the call has the same composition (same kernels, same inputs),
              and within the two-fp16-paths bar of tests/_lgcmp.py when matched alone (the one-pair call runs the latency kernels);
              pair_schedule covers the 28 pairs exactly once for every world size.
  configs[2]  752 x 480 frames, max_kp 600, sharded by blocks: the gathered (desc, kp, n) == the single-process concatenation bit for bit.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_multirank_rehearsal import ROOT, _last_json, _two_ranks

pytestmark = pytest.mark.gpu

CAMS, H, W, K = 8, 720, 1280, 1024


def _single(script_args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, *script_args], capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def test_pair_schedule_covers_every_cross_camera_pair_exactly_once():
    from superslam_amd.shard import pair_schedule

    want = {(i, j) for i in range(CAMS) for j in range(i + 1, CAMS)}
    assert len(want) == 28
    for world in (1, 2, 3, 4, 8):
        sched = pair_schedule(CAMS, world)
        flat = [p for r in sched for p in r]
        assert len(flat) == 28 and set(flat) == want, world
        assert max(map(len, sched)) - min(map(len, sched)) <= 1      # balanced: 28 / 8 -> 3 or 4 per rank


@pytest.fixture(scope="module")
def multicam_runs(tmp_path_factory):
    """scripts/multicam.py at configs[4]'s size: world 1, and two ranks on one GPU (four cameras and fourteen pairs each)."""
    d = tmp_path_factory.mktemp("multicam")
    script = os.path.join(ROOT, "scripts", "multicam.py")
    args = ["--cameras", str(CAMS), "--h", str(H), "--w", str(W), "--max-kp", str(K), "--ticks", "2"]
    out1 = _single([script, *args, "--dump", str(d / "w1")])
    backend, out2 = _two_ranks([script, *args, "--dump", str(d / "w2"), "--gpus", "2"], 29651, plain=True)
    j1, j2 = _last_json(out1), _last_json(out2)
    assert j1["ranks"] == 1 and j1["pairs_total"] == 28 and j1["pairs_this_rank"] == 28
    assert j2["ranks"] == 2 and j2["pairs_this_rank"] == 14
    runs = {1: [np.load(str(d / "w1.rank0.npz"))], 2: [np.load(str(d / f"w2.rank{r}.npz")) for r in range(2)]}
    return runs, backend, (j1, j2)


def test_multicam_cross_camera_matches_equal_direct_matching(multicam_runs, weights_dir, parity_report):
    import torch

    import _lgcmp
    from superslam_amd import LightGlue, SuperPoint
    from superslam_amd.synth import make_frame

    runs, backend, (j1, j2) = multicam_runs
    # --- the direct route: per-camera `extract` (the reference's call, one image at a time), nothing gathered
    sp = SuperPoint(weights_dir["sp_path"], K, 0.005, 4)
    assert sp.initialize(), sp.last_error
    base = make_frame(H, W, 515)
    cams = [np.roll(base, (0, 24 * c), axis=(0, 1)) for c in range(CAMS)]           # scripts/multicam.py's rig
    desc = torch.zeros((CAMS, K, 256), dtype=torch.float16, device="cuda")
    kp = torch.zeros((CAMS, K, 3), dtype=torch.float32, device="cuda")
    n = torch.zeros((CAMS,), dtype=torch.int32, device="cuda")
    feats = []
    lg1 = LightGlue(weights_dir["lg_path"], W, H, max_keypoints=K)
    assert lg1.initialize(), lg1.last_error
    for c in range(CAMS):
        f = sp.extract(cams[c])
        assert len(f.keypoints) == K, len(f.keypoints)                               # the rig saturates max_kp (configs[4]: 1024 per camera)
        feats.append(f)
        kp[c, :K] = torch.from_numpy(np.ascontiguousarray(f.keypoints[:, :3]))
        n[c] = K
        desc[c, :K] = torch.from_numpy(lg1.descriptors_to_host(f.descriptors)).half()  # fp16 -> fp32 -> fp16 is exact
    # every rank of every world size gathered exactly these tensors
    for world, files in runs.items():
        for z in files:
            np.testing.assert_array_equal(z["n"], n.cpu().numpy())
            np.testing.assert_array_equal(z["kp"].view(np.uint32), kp.cpu().numpy().view(np.uint32))
            np.testing.assert_array_equal(z["desc"].view(np.uint16), desc.cpu().numpy().view(np.uint16))
    # --- same composition as the rank's call -> bit for bit
    seen = {}
    for world, files in runs.items():
        covered = []
        for z in files:
            pairs = [tuple(p) for p in z["pairs"]]
            covered += pairs
            lg = LightGlue(lg1.shared_engine(), W, H, max_keypoints=K, max_pairs=len(pairs))
            assert lg.initialize(), lg.last_error
            idx = torch.tensor([c for p in pairs for c in p], dtype=torch.long, device="cuda")
            m0, ms0 = lg.match_batch_device(kp.index_select(0, idx).contiguous(), n.index_select(0, idx).contiguous(),
                                            desc.index_select(0, idx).contiguous())
            torch.cuda.synchronize()
            np.testing.assert_array_equal(z["m0"], m0.cpu().numpy())
            np.testing.assert_array_equal(z["ms0"].view(np.uint32), ms0.cpu().numpy().view(np.uint32))
            for q, p in enumerate(pairs):
                seen.setdefault(p, []).append((z["m0"][q], z["ms0"][q]))
            lg.close()
        assert sorted(covered) == sorted({(i, j) for i in range(CAMS) for j in range(i + 1, CAMS)}), world
    # --- world 1 (one 28-pair call) against world 2 (two 14-pair calls) against each pair matched ALONE from the per-camera handles
    worst = {"agree": 1.0, "maxd": 0.0}
    nmatch = []
    for (i, j), got in sorted(seen.items()):
        r = lg1.match(feats[i].keypoints, feats[i].descriptors, feats[j].keypoints, feats[j].descriptors)
        nmatch.append(int((r.matches0 >= 0).sum()))
        for m0, ms0 in got:
            c = _lgcmp.compare(m0[:K], ms0[:K], r.matches0, r.mscores0, bar=_lgcmp.PATH_VS_PATH_BAR)   # two fp16 paths: the sum rule
            _lgcmp.check(c)
            worst["agree"] = min(worst["agree"], c["agreement"]); worst["maxd"] = max(worst["maxd"], c["mscores_maxd"])
    assert min(nmatch) > 10, nmatch            # overlapping views (24 px apart per camera): every pair has real matches (40-49 measured on the seeded weights)
    print(f"configs[4] at size ({CAMS} x {W}x{H}, {K} kp, 28 pairs; two-rank rehearsal on {backend}): gathered tensors == per-camera extract, "
          f"same-composition calls bit-identical, one-pair calls agree {worst['agree']:.4f} / max|d| {worst['maxd']:.4f}; "
          f"matches per pair {min(nmatch)}..{max(nmatch)}; {j1['ticks_per_s']} ticks/s on one GPU")
    parity_report["config4_multicam_at_size"] = {"bit_identical_same_composition": True, "pairs": 28, "one_pair_agree_min": worst["agree"],
                                                 "mscores_maxd_one_pair_vs_batch": worst["maxd"], "rehearsal_backend": backend,
                                                 "ticks_per_s_1gpu": j1["ticks_per_s"]}
    sp.close(); lg1.close()


def test_offline_extract_at_config_size_two_ranks_equals_single_process(tmp_path, parity_report):
    """configs[2] at 752 x 480, max_kp 600, 64 frames in batches of 32: gathered tensor == single-process concatenation, bit for bit."""
    one, two = str(tmp_path / "one.npz"), str(tmp_path / "two.npz")
    script = os.path.join(ROOT, "scripts", "offline_extract.py")
    args = ["--frames", "64", "--batch", "32", "--h", "480", "--w", "752", "--max-kp", "600"]
    j1 = _last_json(_single([script, *args, "--dump", one]))
    backend, out = _two_ranks([script, *args, "--dump", two, "--gpus", "2"], 29671, plain=True)
    j2 = _last_json(out)
    assert j1["ranks"] == 1 and j2["ranks"] == 2 and j2["frames"] == 64 and j2["pool_bytes"] == 64 * 600 * 256 * 2
    a, b = np.load(one), np.load(two)
    assert a["desc"].shape == (64, 600, 256) and a["kp"].shape == (64, 600, 3)
    np.testing.assert_array_equal(a["n"], b["n"])
    np.testing.assert_array_equal(a["kp"].view(np.uint32), b["kp"].view(np.uint32))
    np.testing.assert_array_equal(a["desc"].view(np.uint16), b["desc"].view(np.uint16))
    assert int(a["n"].min()) == 600                     # the frames saturate max_kp: every descriptor row is live
    # and the batched extraction == the reference's one-image call on a sample of frames (the pool image is what a consumer would read)
    from superslam_amd import LightGlue, SuperPoint
    from superslam_amd.synth import make_frame
    from superslam_amd.weights import make_superpoint_weights, save_safetensors

    save_safetensors(make_superpoint_weights(0), str(tmp_path / "sp.safetensors"))
    sp = SuperPoint(str(tmp_path / "sp.safetensors"), 600, 0.005, 4)
    assert sp.initialize(), sp.last_error
    base = make_frame(480, 752, 4242)
    from superslam_amd import _lib
    import ctypes as C
    for f in (0, 31, 32, 63):
        img = np.roll(base, ((f * 37) % 480, (f * 101) % 752), axis=(0, 1))
        ft = sp.extract(img)
        np.testing.assert_array_equal(ft.keypoints[:, :3].astype(np.float32).view(np.uint32), a["kp"][f].view(np.uint32))
        d = np.zeros((600, 256), np.float32)
        _lib.check(_lib.lib().sship_desc_to_host(ft.descriptors.data, 600, 256, d.ctypes.data))
        np.testing.assert_array_equal(d.astype(np.float16).view(np.uint16), a["desc"][f].view(np.uint16))
    sp.close()
    print(f"configs[2] at size (64 x 752x480, 600 kp; {backend}): gathered == single process bit for bit; {j1['frames_per_s']} frames/s on one GPU")
    parity_report["config2_offline_extract_at_size"] = {"frames": 64, "bit_identical": True, "rehearsal_backend": backend,
                                                        "frames_per_s_1gpu": j1["frames_per_s"]}
