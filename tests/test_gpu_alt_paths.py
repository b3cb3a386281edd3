"""GPU: the A/B kernel paths of the DEVELOPER build still agree with the shipped library's single path, and the measurement probe answers.
The shipped libsuperslam_hip.so has no kernel-selection switches (include/sship.h, "Environment"); a mode with SUPERSLAM_HIP_* switches runs on
superslam_amd/lib/variants/dev.so (superslam_amd/build.py: DEV_SOURCES, -DSSHIP_DEV_SWITCHES=1; built by __graft_entry__.build()), which the
worker selects EXPLICITLY (superslam_amd._lib.set_library_path through scripts/_devlib.py - the product package reads no variable for this);
a mode without switches runs on the shipped library.  The switches are read once per process, so every
mode runs in its own interpreter."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV_LIB = os.path.join(ROOT, "superslam_amd", "lib", "variants", "dev.so")


def _env(mode_env):
    """Process environment of one mode: switches select kernels only in the developer build."""
    env = dict(os.environ, **mode_env)
    env.pop("SSHIP_DEV_LIBRARY", None)
    if mode_env:
        assert os.path.exists(DEV_LIB), "superslam_amd/lib/variants/dev.so is missing: __graft_entry__.build() produces it"
        env["SSHIP_DEV_LIBRARY"] = DEV_LIB      # read by scripts/_devlib.use_dev_library() in the worker, not by the package
    return env


_WORKER = r"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + '/scripts'); import _devlib; _devlib.use_dev_library()
from superslam_amd import SuperPoint, _lib
from superslam_amd.synth import make_stereo_pair
_lib.init(0)
sp = SuperPoint({sp_path!r}, 600, 0.005, 4, max_batch=2); assert sp.initialize(), sp.last_error
l, r = make_stereo_pair(200, 328, 77)
fl, fr = sp.extract_stereo(l, r)
out = {{}}
for tag, f in (("l", fl), ("r", fr)):
    d = np.zeros((f.descriptors.count, 256), np.float32)
    assert _lib.lib().sship_desc_to_host(f.descriptors.data, f.descriptors.count, 256, d.ctypes.data) == 0
    out["kp_" + tag] = f.keypoints; out["d_" + tag] = d.astype(np.float16)
np.savez({out!r}, **out)
"""


def _run(mode_env, weights_dir, tmp_path, name):
    out = str(tmp_path / (name + ".npz"))
    env = _env(mode_env)
    code = _WORKER.format(root=ROOT, sp_path=weights_dir["sp_path"], out=out)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    return np.load(out)


def _ulp16(a, b):
    def key(x):
        u = x.astype(np.float16).view(np.uint16).astype(np.int32)
        return np.where(u & 0x8000, 0x8000 - (u & 0x7FFF), 0x8000 + u)
    return np.abs(key(a) - key(b))


def _assert_same_features_fp16(name, base, alt):
    """Two conv kernels with different fp32 summation orders: same keypoint set (>= 98 % IoU), scores within 2e-2 (the peaky
    synthetic detector amplifies logit differences), descriptors of common keypoints within 1e-2 (unit vectors, fp16)."""
    for tag in ("l", "r"):
        kb, ka = base["kp_" + tag], alt["kp_" + tag]
        ib = {(int(x), int(y)): i for i, (x, y, _) in enumerate(kb)}
        ia = {(int(x), int(y)): i for i, (x, y, _) in enumerate(ka)}
        common = sorted(set(ib) & set(ia))
        iou = len(common) / max(1, len(set(ib) | set(ia)))
        rows_b = np.array([ib[c] for c in common]); rows_a = np.array([ia[c] for c in common])
        ds = np.abs(kb[rows_b, 2] - ka[rows_a, 2]).max()
        dd = np.abs(base["d_" + tag][rows_b].astype(np.float32) - alt["d_" + tag][rows_a].astype(np.float32)).max()
        print(name, tag, "IoU", round(iou, 4), "score max|d|", float(ds), "descriptor max|d|", float(dd))
        assert iou >= 0.98 and ds < 2e-2 and dd < 1e-2


def test_dense_descriptor_branch_agrees_with_sparse_head(weights_dir, tmp_path):
    """SUPERSLAM_HIP_DESC=dense (dense convDa + gather kernel): identical keypoints (same encoder / detector) and
    descriptors within one fp16 ulp of the default head evaluated at the keypoints only (same operands, same k order)."""
    base = _run({}, weights_dir, tmp_path, "default")
    alt = _run({"SUPERSLAM_HIP_DESC": "dense"}, weights_dir, tmp_path, "dense")
    for tag in ("l", "r"):
        np.testing.assert_array_equal(base["kp_" + tag], alt["kp_" + tag])
        d = _ulp16(base["d_" + tag], alt["d_" + tag])
        print("dense head", tag, "n", len(base["kp_" + tag]), "descriptor max ulp", int(d.max()), "exact", float(np.mean(d == 0)))
        assert d.max() <= 1


def test_strip_conv_kernel_agrees_with_ping_pong(weights_dir, tmp_path):
    """SUPERSLAM_HIP_CONV=strip: the lock-step conv kernel adds conv1a's bias in fp32 after the MFMA, the ping-pong
    kernel carries it as an fp16 hi/lo pair in two K-padding slots - activations differ in the last fp16 bits, so the
    comparison is by the suite's fp16 tolerances: same keypoint set (>= 98 % IoU), scores within 2e-2 (the peaky synthetic
    detector amplifies logit differences), descriptors of common keypoints within 1e-2 (unit vectors, fp16)."""
    base = _run({}, weights_dir, tmp_path, "default")
    alt = _run({"SUPERSLAM_HIP_CONV": "strip"}, weights_dir, tmp_path, "strip")
    _assert_same_features_fp16("strip", base, alt)


def test_ct32_conv_kernel_is_bit_identical_to_the_16_row_tile_kernel(weights_dir, tmp_path):
    """The 128-input-channel layers have two kernels on the default path: conv_pp128.hip's 16 x 32-pixel tiles with 64-row cout
    tiles (throughput batches) and conv_pp.hip's 8 x 32-pixel tiles with 32-row cout tiles (latency mode: a frame or two per call
    would leave most CUs without a workgroup on the big tiles; sp_conv3x3_pp picks by tile count).  Both feed the same fp16 operands
    into an fp32 accumulator that starts at the bias IN THE SAME ORDER (32-channel half, kx, k-step, ky), so they are bit-identical:
    a frame extracted alone equals the same frame inside a batch.  SUPERSLAM_HIP_CONV128=th16 / ct32 force one or the other.
    The 16-row kernel packs the narrow right-edge strips of two images into one tile where the shapes allow (SUPERSLAM_HIP_CONV128_PAIRS=0: off):
    the worker's stereo pair has such strips, and the result must not depend on the packing either.
    The worker's 200 x 328 image gives those layers odd tile counts, partial edge tiles and a one-tile-per-group tail."""
    base = _run({"SUPERSLAM_HIP_CONV128": "th16"}, weights_dir, tmp_path, "th16")
    alt = _run({"SUPERSLAM_HIP_CONV128": "ct32"}, weights_dir, tmp_path, "ct32")
    auto = _run({}, weights_dir, tmp_path, "auto")
    # the 16-row kernel's shared edge tiles (the 9-column strips of the pair's two 25 x 41-cell maps in ONE tile; conv4a / 4b / Pa) on and off
    nopairs = _run({"SUPERSLAM_HIP_CONV128": "th16", "SUPERSLAM_HIP_CONV128_PAIRS": "0"}, weights_dir, tmp_path, "th16_nopairs")
    for tag in ("l", "r"):
        for other in (alt, auto, nopairs):
            np.testing.assert_array_equal(base["kp_" + tag], other["kp_" + tag])
            np.testing.assert_array_equal(base["d_" + tag].view(np.uint16), other["d_" + tag].view(np.uint16))


def test_streaming_convpb_agrees_with_the_implicit_gemm_template(weights_dir, tmp_path):
    """SUPERSLAM_HIP_CONVPB=igemm: detector logits through the generic 1x1 kernel (32x32x16 MFMA, LDS-staged) instead of
    k_convpb_stream (16x16x32 MFMA, operands straight from global memory).  Different k grouping inside the matrix
    instructions: compared by the suite's fp16 tolerances."""
    base = _run({}, weights_dir, tmp_path, "default")
    alt = _run({"SUPERSLAM_HIP_CONVPB": "igemm"}, weights_dir, tmp_path, "igemm")
    _assert_same_features_fp16("convPb igemm", base, alt)


_LG_WORKER = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + '/scripts'); import _devlib; _devlib.use_dev_library()
from superslam_amd import LightGlue, _lib
_lib.init(0)
P, K = 64, 600
lg = LightGlue({lg_path!r}, 1376, 376, max_keypoints=K, max_pairs=P); assert lg.initialize(), lg.last_error
g = torch.Generator().manual_seed(5)
kp = (torch.rand((2 * P, K, 3), generator=g) * torch.tensor([1376.0, 376.0, 1.0])).cuda()
ds = torch.nn.functional.normalize(torch.randn((2 * P, K, 256), generator=g), dim=-1).half().cuda()
n = torch.randint(300, K + 1, (2 * P,), generator=g, dtype=torch.int32).cuda()
m, sc = lg.match_batch_device(kp, n, ds)
torch.cuda.synchronize()
np.savez({out!r}, m=m.cpu().numpy(), s=sc.cpu().numpy())
"""


def test_two_stream_lightglue_is_bit_identical_to_one_stream(weights_dir, tmp_path):
    """lg_forward runs the layer stack of a 64-pair batch as two half-batches on two streams (SUPERSLAM_HIP_LG_SPLIT=1: one
    stream).  Pairs are independent and both halves use the same kernels, so matches and scores must be identical bit for
    bit - a missing fork / join edge or an overlapping buffer slice shows up as a difference (ragged counts per sequence)."""
    outs = []
    # (the split path also switches attention to its no-key-split variant - another summation order; pin the variant so that the
    # comparison isolates the stream plumbing)
    for name, env in (("split", {"SUPERSLAM_HIP_ATTN_KS": "2"}), ("nosplit", {"SUPERSLAM_HIP_LG_SPLIT": "1", "SUPERSLAM_HIP_ATTN_KS": "2"})):
        out = str(tmp_path / (name + ".npz"))
        code = _LG_WORKER.format(root=ROOT, lg_path=weights_dir["lg_path"], out=out)
        r = subprocess.run([sys.executable, "-c", code], env=_env(env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    np.testing.assert_array_equal(outs[0]["m"], outs[1]["m"])
    np.testing.assert_array_equal(outs[0]["s"], outs[1]["s"])
    print("two-stream LightGlue: 64 pairs,", int((outs[0]["m"] >= 0).sum()), "matches, identical")


def test_resident_key_attention_is_bit_identical_to_the_streaming_kernel(weights_dir, tmp_path):
    """csrc/lg_attn_res.hip (developer build, SUPERSLAM_HIP_ATTN=res): one workgroup per (sequence, head) with K / V^T filled once into LDS by
    LDS-DMA.  Same MFMAs on the same fragments in the same key order as the shipped k_lg_attention<2, 1, 3> (the kernel a 64-pair call's
    two half-batches run): matches and scores of a ragged 64-pair batch are identical bit for bit - and so is a batch whose sequences
    need the second, single-tile query pass, the staged first sweep and the ragged last key tile (lengths 300 .. 600)."""
    outs = []
    for name, env in (("stream", {}), ("res", {"SUPERSLAM_HIP_ATTN": "res"})):
        out = str(tmp_path / ("attn_" + name + ".npz"))
        code = _LG_WORKER.format(root=ROOT, lg_path=weights_dir["lg_path"], out=out)
        r = subprocess.run([sys.executable, "-c", code], env=_env(env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    np.testing.assert_array_equal(outs[0]["m"], outs[1]["m"])
    np.testing.assert_array_equal(outs[0]["s"], outs[1]["s"])
    print("resident-key attention: 64 pairs,", int((outs[0]["m"] >= 0).sum()), "matches, identical to the streaming kernel")


def test_ffn_kernel_variants_agree(weights_dir, tmp_path, parity_report):
    """The fused FFN block has four kernels: k_lg_ffn4 (throughput batches, default at 64 pairs), the 8-wave k_lg_ffn with 64-token tiles
    (SUPERSLAM_HIP_FFN=8), the same kernel in its latency-mode instantiation (32-token tiles, prefetched projection, LDS-only barriers;
    SUPERSLAM_HIP_FFN_NT=1 forces it onto a 64-pair batch, where every workgroup walks ~9 tiles: the multi-tile path one pair never takes)
    and the 16-wave k_lg_ffn16 (SUPERSLAM_HIP_FFN=16, A/B).  Same operands, same k order per accumulator, fp32 LayerNorm / GELU: compared
    with the default by the path-vs-path bar on ragged sequence lengths; the printout says which variants are bit-identical."""
    import _lgcmp

    res = {}
    for name, env in (("ffn4", {}), ("ffn8", {"SUPERSLAM_HIP_FFN": "8"}), ("ffn8_nt1", {"SUPERSLAM_HIP_FFN": "8", "SUPERSLAM_HIP_FFN_NT": "1"}),
                      ("ffn16", {"SUPERSLAM_HIP_FFN": "16"}), ("noprefetch", {"SUPERSLAM_HIP_LG_PREFETCH": "0"})):
        out = str(tmp_path / ("ffn_" + name + ".npz"))
        code = _LG_WORKER.format(root=ROOT, lg_path=weights_dir["lg_path"], out=out)
        r = subprocess.run([sys.executable, "-c", code], env=_env(env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[name] = np.load(out)
    for name in ("ffn8", "ffn8_nt1", "ffn16", "noprefetch"):
        c = _lgcmp.compare(res[name]["m"].ravel(), res[name]["s"].ravel(), res["ffn4"]["m"].ravel(), res["ffn4"]["s"].ravel(), _lgcmp.PATH_VS_PATH_BAR)
        same = bool(np.array_equal(res[name]["m"], res["ffn4"]["m"]) and np.array_equal(res[name]["s"], res["ffn4"]["s"]))
        print(f"FFN variant {name} vs ffn4: identical {same}, agreement {c['agreement']:.5f}, mscores max|d| {c['mscores_maxd']:.2e}, flips {c['mutual_flips']}")
        _lgcmp.check(c)
        entry = parity_report.setdefault("lg_ffn_kernel_variants", {"pairs": 64, "reference": "k_lg_ffn4"})
        # the default-path pair (ffn4 for batches, the 8-wave kernel for a few pairs) carries the enforced margin; the A/B kernel is recorded only
        entry["mscores_maxd" if name == "ffn8" else name + "_maxd"] = c["mscores_maxd"]
        entry[name + "_agreement"] = c["agreement"]


_DENSE_WORKER = r"""
import sys, numpy as np, torch
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + '/scripts'); import _devlib; _devlib.use_dev_library()
from superslam_amd import SuperPoint, _lib
from superslam_amd.synth import make_frame
_lib.init(0)
out = {{}}
for (h, w, b) in {shapes!r}:
    sp = SuperPoint({sp_path!r}, 600, 0.005, 4, max_batch=b); assert sp.initialize(), sp.last_error
    imgs = torch.from_numpy(np.stack([make_frame(h, w, 31 + i) for i in range(b)])).cuda()
    scores, desc, logits = sp.dense(imgs, want_logits=True)
    torch.cuda.synchronize()
    out["l_%dx%d" % (h, w)] = logits.cpu().numpy(); out["d_%dx%d" % (h, w)] = desc.float().cpu().numpy()
    sp.close()
np.savez({out!r}, **out)
"""
# tile-count corner cases of the ping-pong conv kernels: one partial tile per layer (64 x 64: the conv4 maps are 8 x 8), a single
# tile column, widths that are not multiples of 32 at any pyramid level, odd batch (tiles split unevenly between the wave groups)
_DENSE_SHAPES = [(64, 64, 1), (72, 136, 3), (104, 520, 1), (376, 248, 2)]


def test_conv_kernels_agree_on_tile_corner_cases(weights_dir, tmp_path):
    """Dense logits / descriptor grids of the default conv kernels against the lock-step strip kernel (an independent
    implementation with 64-bit addressing and per-unit masks) and against the 32-row-tile kernel, on image sizes chosen for
    the staging corner cases of conv_pp.hip / conv_pp128.hip (affine buffer addressing, out-of-range zero fill, scalar tile
    walk, uneven tile split between the two wave groups).  fp16 activations, different fp32 summation orders: logits
    (O(25)) within 4e-2, unit-norm descriptor grid within 1e-2."""
    res = {}
    # th8: the 8-row-tile register-staged kernel the 128-channel layers ran on before the 16-row LDS-DMA kernel; dma64: the
    # LDS-DMA kernel also for the 64-channel layers (two resident chunks) - both opt-in A/B paths of conv_pp128.hip
    # th16 / ct32: the two default-path kernels of the 128-input-channel layers, each forced for every shape (bit-identical to each other)
    variants = (("default", {}), ("strip", {"SUPERSLAM_HIP_CONV": "strip"}), ("ct32", {"SUPERSLAM_HIP_CONV128": "ct32"}),
                ("th16", {"SUPERSLAM_HIP_CONV128": "th16"}),
                ("th8", {"SUPERSLAM_HIP_CONV128": "th8"}), ("dma64", {"SUPERSLAM_HIP_CONV64": "dma"}))
    for name, env in variants:
        out = str(tmp_path / ("dense_" + name + ".npz"))
        code = _DENSE_WORKER.format(root=ROOT, sp_path=weights_dir["sp_path"], shapes=_DENSE_SHAPES, out=out)
        r = subprocess.run([sys.executable, "-c", code], env=_env(env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        res[name] = np.load(out)
    for (h, w, _) in _DENSE_SHAPES:
        for alt in ("strip", "ct32", "th8", "dma64"):
            dl = np.abs(res["default"]["l_%dx%d" % (h, w)] - res[alt]["l_%dx%d" % (h, w)]).max()
            dd = np.abs(res["default"]["d_%dx%d" % (h, w)] - res[alt]["d_%dx%d" % (h, w)]).max()
            print(f"{h}x{w} default vs {alt}: logits max|d| {dl:.3e}, descriptor grid max|d| {dd:.3e}")
            assert dl < 4e-2 and dd < 1e-2, (h, w, alt, dl, dd)
        for a in ("default", "ct32"):  # same accumulation order: exactly equal
            np.testing.assert_array_equal(res["th16"]["l_%dx%d" % (h, w)], res[a]["l_%dx%d" % (h, w)])
            np.testing.assert_array_equal(res["th16"]["d_%dx%d" % (h, w)], res[a]["d_%dx%d" % (h, w)])


def test_mfma_probe_reports_a_plausible_rate():
    import ctypes as C

    from superslam_amd import _lib

    _lib.init(0)
    for rnd in (0, 1):
        tf = C.c_float(0)
        assert _lib.lib().sship_mfma_probe(rnd, C.byref(tf)) == 0
        print("mfma probe", "random" if rnd else "zero", "operands:", round(tf.value, 1), "TFLOP/s")
        assert 200.0 < tf.value < 3000.0
    assert _lib.lib().sship_mfma_probe(0, None) != 0   # null argument is an error, not a crash


def test_two_matchers_on_shared_weights_run_concurrently(weights_dir):
    """The reference's threading model (SuperSLAM.cc:129-133): the tracking thread's LightGlue (device descriptors)
    and the loop-closure worker's LightGlue built from shared_engine() (host descriptors) match at the same time.
    Handles own their streams and workspaces, the weights are immutable: results must equal the serial ones."""
    import threading

    from superslam_amd import LightGlue, SuperPoint, _lib
    from superslam_amd.synth import make_stereo_pair

    _lib.init(0)
    sp = SuperPoint(weights_dir["sp_path"], 600, 0.005, 4, max_batch=2); assert sp.initialize(), sp.last_error
    lg_a = LightGlue(weights_dir["lg_path"], 328, 200, max_keypoints=600); assert lg_a.initialize(), lg_a.last_error
    lg_b = LightGlue(lg_a.shared_engine(), 328, 200, max_keypoints=600); assert lg_b.initialize(), lg_b.last_error
    l, r = make_stereo_pair(200, 328, 31)
    fl, fr = sp.extract_stereo(l, r)
    hl, hr = lg_a.descriptors_to_host(fl.descriptors), lg_a.descriptors_to_host(fr.descriptors)
    ref_a = lg_a.match(fl.keypoints, fl.descriptors, fr.keypoints, fr.descriptors)      # device overload
    ref_b = lg_b.match(fr.keypoints, hr, fl.keypoints, hl)                               # host overload, other direction
    out = {}

    def work(tag, fn, n):
        res = []
        for _ in range(n):
            res.append(fn())
        out[tag] = res

    ta = threading.Thread(target=work, args=("a", lambda: lg_a.match(fl.keypoints, fl.descriptors, fr.keypoints, fr.descriptors), 20))
    tb = threading.Thread(target=work, args=("b", lambda: lg_b.match(fr.keypoints, hr, fl.keypoints, hl), 20))
    ta.start(); tb.start(); ta.join(); tb.join()
    for res in out["a"]:
        np.testing.assert_array_equal(res.matches0, ref_a.matches0)
        np.testing.assert_array_equal(res.mscores0, ref_a.mscores0)
    for res in out["b"]:
        np.testing.assert_array_equal(res.matches0, ref_b.matches0)
        np.testing.assert_array_equal(res.mscores0, ref_b.mscores0)
    print("concurrent matchers: a", int((ref_a.matches0 >= 0).sum()), "matches, b", int((ref_b.matches0 >= 0).sum()))
    del fl, fr
    lg_a.close(); lg_b.close(); sp.close()


def test_winograd_conv2a_conv2b_layer_parity():
    """SUPERSLAM_HIP_CONV64=wino (csrc/conv_wino.hip; A/B, not the default): conv2a and conv2b + pool as Winograd F(2x2, 3x3), each compared
    through the test-only sship_sp_debug_activation with a CPU convolution (fp32 math) of the SAME run's previous activation - a layer
    test, not an end-to-end one.  fp16 transforms cost about 3x the direct kernel's rounding error (direct: 1e-3 on O(3) activations;
    Winograd: 2-3.4e-3); bar 5e-3.  Sizes: two tiles per row with a partial one, an odd width, and the headline frame."""
    import re
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for size in ("64x96", "96x249", "376x1376"):
        env = _env({"SUPERSLAM_HIP_CONV64": "wino"})
        r = subprocess.run([sys.executable, os.path.join(root, "scripts", "dev", "wino_check.py"), size, "2"], capture_output=True, text=True,
                           timeout=600, env=env, cwd=root)
        assert r.returncode == 0, r.stderr[-2000:]
        vals = dict(re.findall(r"wino\s+(conv2a|conv2b\+pool): max\|d\| ([0-9.e+-]+)", r.stdout))
        assert set(vals) == {"conv2a", "conv2b+pool"}, r.stdout
        assert "finite True" in r.stdout and "non-finite" not in r.stdout
        for k, v in vals.items():
            assert float(v) < 5e-3, (size, k, v)


_EP_WORKER = r"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + '/scripts'); import _devlib; _devlib.use_dev_library()
from superslam_amd import _lib
from superslam_amd.synth import make_frame
from superslam_amd.weights import make_eigenplaces_weights, save_safetensors
_lib.init(0); L = _lib.lib()
path = {out!r} + '.safetensors'; save_safetensors(make_eigenplaces_weights(2), path)
res = {{}}
for (w, h) in ((512, 512), (320, 240), (72, 40), (34, 50)):
    hd = C.c_void_p(); _lib.check(L.sship_ep_create(path.encode(), w, h, C.byref(hd)))
    for seed in (3, 4):
        img = make_frame(376, 1241, seed); d = np.zeros(512, np.float32)
        _lib.check(L.sship_ep_infer_u8(hd, img.ctypes.data, 376, 1241, 1241, 1, d.ctypes.data))
        res['%dx%d_%d' % (w, h, seed)] = d
    L.sship_ep_destroy(hd)
np.savez({out!r}, **res)
"""


def test_fused_eigenplaces_stem_agrees_with_the_gemm_stem(tmp_path):
    """k_ep_stem_pool (round 6: 7x7 / stride-2 stem + ReLU + max-pool in one kernel, the patch matrix built in LDS) against the stem of rounds 3-5
    (im2col -> 1x1 GEMM -> max-pool; developer build, SUPERSLAM_HIP_EP_STEM=gemm).  Same fp16 operands, same rounding points (fp32 accumulate,
    bias, ReLU, fp16, max); the k order of the accumulation differs ((c, ky, kx) in steps of 16 over kx padded to 8 against the flat 147-tap row),
    so the descriptors agree to fp32 reassociation through 17 fp16 layers, not bit for bit.  Sizes: the benchmarked 512 x 512, a non-square
    size with partial pooled tiles, one whose pooled map is smaller than one 8 x 8 tile, and odd extents (odd stem / pooled map sizes)."""
    outs = {}
    for name, env in (("fused", {}), ("gemm", {"SUPERSLAM_HIP_EP_STEM": "gemm"})):
        out = str(tmp_path / ("ep_" + name + ".npz"))
        code = _EP_WORKER.format(root=ROOT, out=out)
        r = subprocess.run([sys.executable, "-c", code], env=_env(env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[name] = np.load(out)
    for k in outs["fused"].files:
        a, b = outs["fused"][k], outs["gemm"][k]
        dmax, cos = float(np.abs(a - b).max()), float(a @ b)
        print("EigenPlaces stem fused vs gemm", k, "max|d|", dmax, "cosine", cos)
        assert np.isfinite(a).all() and dmax <= 1e-3 and cos >= 0.99999, (k, dmax, cos)


def test_fused_conv2a_conv2b_is_bit_identical_to_the_two_launches(tmp_path):
    """csrc/conv_fuse2.hip (round 6): conv2a -> conv2b -> max-pool as one rolling-window kernel whose intermediate map never leaves the CU, against the
    two conv3x3_pp launches (developer build, SUPERSLAM_HIP_CONV2=fused / split forces either on every batch size).  Same fp16 operands, same k order
    per accumulator, same rounding points: the pooled conv2b map and the conv3a map behind it are equal BIT FOR BIT.  The conv3a map is also the A/B of
    the kernel's one-layer form (conv_roll<true>: conv3a's 128 output channels in one launch) against conv3x3_pp<64, 64> with its two cout tiles, which
    the library runs; the "fused" mode selects the one-layer form with SUPERSLAM_HIP_CONV3A=roll (measured at the same joules, so not shipped).  Sizes: one and several 30-column
    strips, a partial last strip (widths 164, 160, 620, 125, 75, 48, 32, 688, 960, 20), the engine's maximum frame (1080 x 1920) and a 32 x 40 one, odd half-resolution widths and heights (62 x 125, 185 x 75: the floor
    pooling drops the last row / column), strips cut into two row segments and not, heights that are not multiples of the 4-row step, 2-4 images."""
    outs = {}
    # "fused" cuts every strip into the row segments the library picks for the batch (conv_fuse2.hip: f2_nseg); the two forced segment counts prove
    # that the cut never changes the arithmetic: one segment per strip (the longest pipelines) and seven (segments of 2-7 steps, many seams)
    for mode, env in (("split", {"SUPERSLAM_HIP_CONV2": "split"}), ("fused", {"SUPERSLAM_HIP_CONV2": "fused", "SUPERSLAM_HIP_CONV3A": "roll"}),
                      ("fused_nseg1", {"SUPERSLAM_HIP_CONV2": "fused", "SUPERSLAM_HIP_CONV2_NSEG": "1"}),
                      ("fused_nseg7", {"SUPERSLAM_HIP_CONV2": "fused", "SUPERSLAM_HIP_CONV2_NSEG": "7"})):
        out = str(tmp_path / ("conv2_" + mode + ".npz"))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "dev", "fuse2_dump.py"), out], env=_env(env),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = np.load(out)
    assert len(outs["split"].files) >= 16
    for mode in ("fused", "fused_nseg1", "fused_nseg7"):
        for k in outs["split"].files:
            a, b = outs["split"][k], outs[mode][k]
            nz = int((a != b).sum())
            print("conv2", mode, "vs split", k, a.shape, "differing halfs:", nz)
            assert nz == 0, (mode, k, nz, np.argwhere(a != b)[:8].tolist())
            assert a.any()
