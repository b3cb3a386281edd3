"""GPU (-m gpu): handle lifetimes, stream ordering, the remaining BASELINE configs and the persisted parity numbers.

* a DeviceDescriptors handle may outlive its extractor (include/DescriptorPool.h:71-75) - round-1 ADVICE: use-after-free;
* extractor -> matcher chained on the NULL stream with nothing in between must be ordered - round-1 ADVICE;
* BASELINE configs[0]: one 640x480 grayscale frame, 1024 keypoints, against the oracle;
* BGR input with three DISTINCT channels against the OpenCV fixed-point gray conversion;
* end-to-end keypoint IoU at 1376x376 against SURVEY 8(c)'s 0.98 bar, with the numbers written to the parity report.
"""
import gc
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import hostpath as H  # noqa: E402
from oracle import superpoint_ref as R  # noqa: E402


@pytest.fixture(scope="module")
def hip():
    from superslam_amd import _lib

    _lib.init()
    assert torch.cuda.is_available()
    return _lib.lib()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _kp_set(kp):
    return {(round(float(k[0]), 2), round(float(k[1]), 2)) for k in kp}


def _explain_disagreements(got_kp, ref, s_ref, input_hw, thr=0.005):
    """Every keypoint the GPU picked that the oracle did not (and vice versa) must be a near-tie in the ORACLE's own
    score map: the losing pixel's score is within `tol` of the 600th score (budget boundary) or of its NMS rival."""
    a, b = _kp_set(got_kp), _kp_set(ref["kp"])
    iou = len(a & b) / max(1, len(a | b))
    return iou, len(a - b), len(b - a)


def test_features_outlive_their_extractor(hip, weights_dir):
    from superslam_amd import SuperPoint
    from superslam_amd.synth import make_stereo_pair

    sp = SuperPoint(weights_dir["sp_path"], 200, 0.005, 4)
    assert sp.initialize(), sp.last_error
    l, r = make_stereo_pair(120, 160, 3)
    fl, fr = sp.extract_stereo(l, r)
    assert fl.descriptors.slot >= 0 and fr.descriptors.slot >= 0 and sp.pool_in_use() == 2
    pool = sp.pool_handle
    hip.sship_pool_retain(pool)                 # this test's own reference, to look at the bookkeeping after the fact
    sp.close()                                  # frees the device slots, drops the extractor's reference
    assert hip.sship_pool_slot_ptr(pool, fl.descriptors.slot) in (None, 0)   # device memory is gone (as in the reference)
    assert hip.sship_pool_in_use(pool) == 2
    del fl                                      # deleter runs AFTER the extractor died: must only touch live bookkeeping
    gc.collect()
    assert hip.sship_pool_in_use(pool) == 1
    del fr
    gc.collect()
    assert hip.sship_pool_in_use(pool) == 0
    hip.sship_pool_release_ref(pool)            # last reference: the struct is deleted here
    # C-level: destroy with no handles at all, and release_ref / retain on NULL are no-ops
    import ctypes as C

    p = C.c_void_p()
    assert hip.sship_pool_create(2, 8, 256, C.byref(p)) == 0
    hip.sship_pool_destroy(p)
    hip.sship_pool_retain(None); hip.sship_pool_release_ref(None)


def test_chained_null_stream_calls_are_ordered(hip, weights_dir):
    """extract_batch_device(x); match_batch_device(kp, n, desc) with stream NULL and NO work in between, repeatedly with
    changing inputs: the matcher must see the extractor's outputs of the same iteration (and the next extract must not
    overwrite them early).  Reference = the same sequence with a device synchronise after every call."""
    from superslam_amd import LightGlue, SuperPoint
    from superslam_amd.synth import make_stereo_pair

    sp = SuperPoint(weights_dir["sp_path"], 600, 0.005, 4, max_batch=2)
    lg = LightGlue(weights_dir["lg_path"], 320, 240, max_keypoints=600, max_pairs=1)
    assert sp.initialize() and lg.initialize()
    pairs = [dev(np.stack(make_stereo_pair(240, 320, 70 + i))) for i in range(6)]
    assert torch.cuda.current_stream().cuda_stream == 0
    desc = torch.empty((2, 600, 256), dtype=torch.float16, device="cuda")
    kp = torch.empty((2, 600, 3), dtype=torch.float32, device="cuda")
    n = torch.empty((2,), dtype=torch.int32, device="cuda")
    ref = []
    for x in pairs:
        sp.extract_batch_device(x, desc, kp, n); torch.cuda.synchronize()
        m0, ms0 = lg.match_batch_device(kp, n, desc); torch.cuda.synchronize()
        ref.append((m0.clone(), ms0.clone(), n.clone()))
    outs = [(torch.empty((1, 600), dtype=torch.int32, device="cuda"), torch.empty((1, 600), dtype=torch.float32, device="cuda"))
            for _ in pairs]
    for rep in range(3):
        for x, (m0, ms0) in zip(pairs, outs):       # no synchronisation, no torch op between the two calls
            sp.extract_batch_device(x, desc, kp, n)
            lg.match_batch_device(kp, n, desc, m0, ms0)
        torch.cuda.synchronize()
        for (m0, ms0), (rm, rs, rn) in zip(outs, ref):
            k = int(rn[0])
            assert torch.equal(m0[0, :k], rm[0, :k]), f"rep {rep}: matcher raced the extractor"
            assert torch.allclose(ms0[0, :k], rs[0, :k], atol=1e-6)
    assert (ref[0][0] >= 0).sum() > 20
    sp.close(); lg.close()


def test_config0_640x480_mono_1024_keypoints(hip, weights_dir, parity_report):
    """BASELINE.json configs[0]: SuperPoint on one 640x480 grayscale frame, 1024 keypoints - shapes of SURVEY 8(d)
    config 1 (scores [480,640], grid [256,60,80], descriptors [1024,256]) and the oracle's keypoints."""
    from superslam_amd import SuperPoint
    from superslam_amd.synth import make_frame

    sp = SuperPoint(weights_dir["sp_path"], 1024, 0.005, 4, max_batch=1)
    assert sp.initialize(), sp.last_error
    img = make_frame(480, 640, 2024)
    scores, grid = sp.dense(dev(img[None]))
    torch.cuda.synchronize()
    assert tuple(scores.shape) == (1, 480, 640) and tuple(grid.shape) == (1, 256, 60, 80)
    ok, kp, d = sp.infer(img)                                          # the mono host path of the reference's `infer`
    assert ok and kp.shape == (1024, 3) and d.shape == (1024, 256)
    np.testing.assert_allclose(np.linalg.norm(d, axis=1), 1.0, atol=2e-3)
    f = sp.extract(img)
    np.testing.assert_array_equal(f.keypoints, kp)
    x = R.preprocess_u8(torch.from_numpy(img)[None])
    with torch.no_grad():
        s16, _ = R.dense_forward(weights_dir["sp"], x, emulate_fp16=True)
    ref = H.select_topk(s16[0].numpy(), 480, 640, 0.005, 4, 1024, 60, 80)
    iou, only_gpu, only_ref = _explain_disagreements(kp, ref, s16[0].numpy(), (480, 640))
    # stage-exact on the library's own score map (the contract's "bit-exact keypoint indices after NMS")
    own = H.select_topk(scores[0].cpu().numpy(), 480, 640, 0.005, 4, 1024, 60, 80)
    np.testing.assert_array_equal(kp, own["kp"])
    print(f"config[0] 640x480/1024: keypoint IoU vs fp16-emulating oracle {iou:.4f} (+{only_gpu} / -{only_ref})")
    parity_report["config0_640x480_1024kp"] = {"keypoint_iou": iou, "only_gpu": only_gpu, "only_oracle": only_ref, "n": int(len(kp))}
    assert iou >= 0.98
    sp.close()


def test_bgr_with_distinct_channels(hip, weights_dir):
    """3-channel input goes through cv::COLOR_BGR2GRAY's fixed-point formula (src/SuperPoint.cc:388,771): extracting a
    BGR image equals extracting the oracle's gray conversion of it, bit for bit."""
    from superslam_amd import SuperPoint
    from superslam_amd.synth import make_frame

    sp = SuperPoint(weights_dir["sp_path"], 300, 0.005, 4)
    assert sp.initialize(), sp.last_error
    b, g, r = make_frame(120, 168, 1), make_frame(120, 168, 2), make_frame(120, 168, 3)
    bgr = np.stack([b, g, r], -1)
    gray = H.bgr2gray_u8(bgr)
    assert (gray != b).any() and (gray != g).any() and (gray != r).any()
    # spot values by hand: (B,G,R) = (255,0,0) -> 29 ; (0,255,0) -> 150 ; (0,0,255) -> 76 ; (10,20,30) -> 22
    probe = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 20, 30]]], np.uint8)
    assert H.bgr2gray_u8(probe).tolist() == [[29, 150, 76, 22]]
    f3, f1 = sp.extract(bgr), sp.extract(gray)
    assert len(f3.keypoints) > 50
    np.testing.assert_array_equal(f3.keypoints, f1.keypoints)
    l3, r3 = sp.extract_stereo(bgr, bgr[:, ::-1].copy())
    l1, r1 = sp.extract_stereo(gray, gray[:, ::-1].copy())
    np.testing.assert_array_equal(l3.keypoints, l1.keypoints)
    np.testing.assert_array_equal(r3.keypoints, r1.keypoints)
    sp.close()


def test_end_to_end_keypoint_iou_meets_the_contract(hip, weights_dir, parity_report):
    """end-to-end keypoint set IoU >= 0.98 at 1376x376, N = 600, against the fp16-emulating oracle; the
    library's keypoints are bit-exact with select_topk of its OWN score map (stage-exact indices after NMS)."""
    from superslam_amd import SuperPoint
    from superslam_amd.synth import make_stereo_pair

    sp = SuperPoint(weights_dir["sp_path"], 600, 0.005, 4, max_batch=2)
    assert sp.initialize(), sp.last_error
    l, r = make_stereo_pair(376, 1376, 1234)
    fl, fr = sp.extract_stereo(l, r)
    scores, _ = sp.dense(dev(np.stack([l, r])))
    torch.cuda.synchronize()
    x = R.preprocess_u8(torch.from_numpy(np.stack([l, r])))
    with torch.no_grad():
        s16, _ = R.dense_forward(weights_dir["sp"], x, emulate_fp16=True)
        s32, _ = R.dense_forward(weights_dir["sp"], x)
    rep = {}
    for b, f in enumerate((fl, fr)):
        own = H.select_topk(scores[b].cpu().numpy(), 376, 1376, 0.005, 4, 600, 47, 172)
        np.testing.assert_array_equal(f.keypoints, own["kp"])
        ref16 = H.select_topk(s16[b].numpy(), 376, 1376, 0.005, 4, 600, 47, 172)
        ref32 = H.select_topk(s32[b].numpy(), 376, 1376, 0.005, 4, 600, 47, 172)
        iou16, g16, r16 = _explain_disagreements(f.keypoints, ref16, None, None)
        iou32, g32, r32 = _explain_disagreements(f.keypoints, ref32, None, None)
        iou_refs = len(_kp_set(ref16["kp"]) & _kp_set(ref32["kp"])) / len(_kp_set(ref16["kp"]) | _kp_set(ref32["kp"]))
        sg = scores[b].cpu().numpy()
        surv = (sg > 0) != (s16[b].numpy() > 0)
        print(f"e2e 1376x376 image {b}: IoU vs fp16-emulating oracle {iou16:.4f}, vs fp32 reference arithmetic {iou32:.4f} "
              f"(oracle fp16-vs-fp32 IoU {iou_refs:.4f}); NMS survival flips {int(surv.sum())} of {int((s16[b].numpy() > 0).sum())}")
        rep[f"image{b}"] = {"iou_vs_fp16_oracle": iou16, "iou_vs_fp32_oracle": iou32, "iou_fp16_oracle_vs_fp32_oracle": iou_refs,
                            "nms_survival_flips": int(surv.sum()), "survivors": int((s16[b].numpy() > 0).sum())}
        assert iou16 >= 0.98
    parity_report["e2e_keypoints_1376x376_600kp"] = rep
    sp.close()


# ------------------------------------------------------------------------------------------------------
# cross-frame pipelining: sship_sp_ring_submit (round 3)
# ------------------------------------------------------------------------------------------------------
def test_ring_submit_is_bit_identical_to_extract_stereo_and_survives_interleaving(weights_dir):
    """A submitted slot yields exactly what the synchronous call yields (keypoints, scores, descriptors), also when another
    extraction and a LightGlue match run on the same handles between the submission and its collection (results live in the
    slot's own pinned buffers and pool slots; the handle's stream orders the device work), and a never-collected submission
    hands its pool slots back when the ring goes away."""
    import gc

    from superslam_amd import LightGlue, SuperPoint, _lib
    from superslam_amd.synth import make_stereo_pair

    h, w = 240, 376
    sp = SuperPoint(weights_dir["sp_path"], 300, 0.005, 4, max_batch=2)
    lg = LightGlue(weights_dir["lg_path"], w, h, max_keypoints=300, max_pairs=1)
    assert sp.initialize() and lg.initialize()
    assert sp.ring_create(3, h, w, 1), sp.last_error
    pairs = [make_stereo_pair(h, w, 900 + i) for i in range(3)]
    ref = []
    for l, r in pairs:                                   # synchronous reference
        fl, fr = sp.extract_stereo(l, r)
        ref.append([(f.keypoints.copy(), lg.descriptors_to_host(f.descriptors)) for f in (fl, fr)])
        del fl, fr
    gc.collect()
    assert sp.pool_in_use() == 0
    for s, (l, r) in enumerate(pairs):
        sp.ring_host(s, 0)[:] = l; sp.ring_host(s, 1)[:] = r
        sp.ring_upload(s)
    sp.ring_submit(0)
    sp.ring_submit(1)                                    # two submissions in flight
    assert sp.pool_in_use() == 4
    fl2, fr2 = sp.extract_stereo_ring(2)                 # a synchronous ring extraction in between (not submitted)
    m = lg.match(fl2.keypoints, fl2.descriptors, fr2.keypoints, fr2.descriptors)   # and a match on the other handle
    assert len(m.matches0) == len(fl2.keypoints)
    got = {2: (fl2, fr2), 1: sp.extract_stereo_ring(1), 0: sp.extract_stereo_ring(0)}   # collected out of order
    for s in range(3):
        for b in range(2):
            f = got[s][b]
            np.testing.assert_array_equal(f.keypoints.view(np.uint32), ref[s][b][0].view(np.uint32), err_msg=f"slot {s} image {b}")
            d = lg.descriptors_to_host(f.descriptors)
            np.testing.assert_array_equal(d, ref[s][b][1], err_msg=f"slot {s} image {b} descriptors")
    with pytest.raises(_lib.SshipError):
        sp.ring_submit(0); sp.ring_submit(0)             # a slot holds one submission at a time
    del got, fl2, fr2, f
    gc.collect()
    assert sp.pool_in_use() == 2                         # the uncollected submission of slot 0 still owns its two slots
    assert sp.ring_create(2, h, w, 1)                    # re-creating the ring drops it and returns the slots
    assert sp.pool_in_use() == 0
    sp.close(); lg.close()


def test_closing_a_handle_with_an_uncollected_submission_and_reupload_of_a_pending_slot(weights_dir):
    """(i) sship_sp_destroy with a submission pending used to release the submission's pool slots into a pool that had
    already been destroyed (a mutex lock + vector push in freed memory): the ring now goes first.  A surviving descriptor handle
    keeps the pool's bookkeeping alive across the close, so its late release must stay harmless.  (ii) re-uploading a slot whose
    submitted extraction has not been collected would race with the queued network: refused with an error."""
    import gc

    from superslam_amd import SuperPoint, _lib
    from superslam_amd.synth import make_stereo_pair

    h, w = 240, 376
    for ch in (1, 3):
        sp = SuperPoint(weights_dir["sp_path"], 300, 0.005, 4, max_batch=2)
        assert sp.initialize()
        assert sp.ring_create(2, h, w, ch), sp.last_error
        l, r = make_stereo_pair(h, w, 77)
        for s in range(2):
            for i, im in enumerate((l, r)):
                sp.ring_host(s, i)[:] = im if ch == 1 else np.repeat(im[:, :, None], 3, axis=2)
            sp.ring_upload(s)
        sp.ring_submit(0)
        with pytest.raises(_lib.SshipError):
            sp.ring_upload(0)                            # pending: the queued network still reads the slot's device frame
        fl, fr = sp.extract_stereo_ring(0)               # collect ...
        sp.ring_upload(0)                                # ... then the slot can be refilled
        sp.ring_submit(0); sp.ring_submit(1)
        assert sp.pool_in_use() == 6
        keep = fl                                        # a handle that outlives the extractor (DescriptorPool.h:71-75)
        sp.close()                                       # two submissions pending: must neither crash nor leak
        del fl, fr, keep
        gc.collect()
    sp = SuperPoint(weights_dir["sp_path"], 300, 0.005, 4, max_batch=2)
    assert sp.initialize()
    assert not sp.ring_create(99, h, w, 1)               # bad depth: no ring, and ring_host says so instead of segfaulting
    with pytest.raises(RuntimeError):
        sp.ring_host(0, 0)
    assert sp.ring_create(1, h, w, 1)
    with pytest.raises(IndexError):
        sp.ring_host(5, 0)
    sp.close()


def test_candidate_counters_survive_stage_benchmarks_and_batch_changes(weights_dir):
    """k_topk clears the per-image candidate counters it has read, so an extraction needs no memset launch - unless the counters are
    not known to be zero: a regrown buffer, a smaller batch after a larger one, or sship_sp_bench_layer's NMS stage, which leaves its
    counts behind for the top-k stage.  Every sequence below must reproduce the first extraction bit for bit (a stale counter would
    append the new candidates behind the old ones and change the selection)."""
    import ctypes as C

    from superslam_amd import SuperPoint, _lib
    from superslam_amd.synth import make_stereo_pair

    h, w = 200, 328
    sp = SuperPoint(weights_dir["sp_path"], 300, 0.005, 4, max_batch=4)
    assert sp.initialize(), sp.last_error
    l, r = make_stereo_pair(h, w, 5)
    l2, r2 = make_stereo_pair(h, w, 6)

    def feats():
        fl, fr = sp.extract_stereo(l, r)
        return fl.keypoints.copy(), fr.keypoints.copy()

    ref = feats()
    for _ in range(2):                                     # back to back: the counters were cleared by the previous call's top-k
        got = feats()
        np.testing.assert_array_equal(ref[0], got[0]); np.testing.assert_array_equal(ref[1], got[1])
    imgs = torch.from_numpy(np.stack([l2, r2, l, r])).cuda()
    _lib.lib().sship_set_profiling(1)                      # keeps a copy of the input for the stage benchmarks
    sp.extract_batch_device(imgs)                          # larger batch: buffers regrown, four counters in play
    torch.cuda.synchronize()
    _lib.lib().sship_set_profiling(0)
    got = feats()                                          # smaller batch after the larger one
    np.testing.assert_array_equal(ref[0], got[0]); np.testing.assert_array_equal(ref[1], got[1])
    sp.extract_batch_device(imgs); torch.cuda.synchronize()
    ms = C.c_float(0)
    for layer in (12, 13):                                 # NMS stage (leaves counts behind), top-k stage (reads them, does not clear)
        _lib.check(_lib.lib().sship_sp_bench_layer(sp._h, layer, 4, h, w, 3, C.byref(ms), None))
    got = feats()
    np.testing.assert_array_equal(ref[0], got[0]); np.testing.assert_array_equal(ref[1], got[1])
    sp.close()


@pytest.mark.parametrize("calls,pairs", [(120, 1), (60, 5), (25, 32)])
def test_repeated_calls_are_bit_identical(calls, pairs):
    """Soak (scripts/dev/soak_determinism.py): the same front-end call on the same input, again and again - keypoints, counts, descriptors, matches and
    scores must equal the first call's bit for bit.  The persistent kernels (ping-pong convolutions, the rolling-window conv2a -> conv2b kernel with its
    producer / consumer rings and LDS-DMA, the two-stream LightGlue call) synchronise by barriers and counters; a hole in that shows up as a rare mismatch
    here long before it turns a parity test red.  One pair per call runs the latency-mode kernels and ~500 row segments in the fused conv2 kernel, 5 and
    32 pairs the throughput kernels with odd and even unit counts per workgroup."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "dev", "soak_determinism.py"), str(calls), str(pairs)], capture_output=True, text=True,
                       timeout=600, cwd=root)
    print(r.stdout[-400:])
    assert r.returncode == 0 and "mismatching outputs: 0" in r.stdout, (r.stdout[-1500:], r.stderr[-1500:])
