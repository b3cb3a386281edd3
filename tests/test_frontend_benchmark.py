"""examples/frontend_benchmark.cc (the reference's per-frame benchmark runner on the HIP
front-end, with the tracker's second LightGlue call): builds on CPU, runs on the GPU box on synthetic pairs and on a
PGM sequence directory."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "superslam_amd", "lib", "frontend_benchmark")


def _build():
    from _cppbuild import LIBDIR, cpp_binary

    # the per-frame benchmark runner is an example PRODUCT binary (examples/): it lives next to the library
    return cpp_binary("frontend_benchmark", [os.path.join(ROOT, "examples", "frontend_benchmark.cc")],
                      deps=[os.path.join(ROOT, "include", "superslam_hip", "frontend.hpp"), os.path.join(ROOT, "include", "superslam_hip", "image_io.hpp")],
                      extra=["-lpthread", "-lz"], opt="-O2", outdir=LIBDIR)


def test_benchmark_runner_builds_and_rejects_bad_usage():
    from superslam_amd import _lib

    _lib.lib()   # the .so must exist (built by __graft_entry__.build / superslam_amd.build)
    out = subprocess.run([_build()], capture_output=True, text=True, timeout=60)
    assert out.returncode == 2 and "usage:" in out.stderr


def _field(text, label):
    m = re.search(label + (r"([0-9.]+)" if label.endswith("=") else r"\s*:\s*([0-9.]+)"), text)
    assert m, text
    return float(m.group(1))


@pytest.mark.gpu
def test_benchmark_runner_synthetic_with_keyframe_match(weights_dir):
    out = subprocess.run([_build(), "--sp", weights_dir["sp_path"], "--lg", weights_dir["lg_path"], "--synthetic", "12",
                          "--keyframe-match"], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert _field(out.stdout, "frames") == 12
    m = re.search(r"stereo matches\s*:\s*([0-9.]+) per frame, ([0-9.]+) pass", out.stdout)
    assert m and float(m.group(1)) > 0 and float(m.group(2)) > 0      # shifted right images: real, gated stereo matches
    assert "keyframe matches" in out.stdout and "per-frame ms" in out.stdout


@pytest.mark.gpu
def test_cross_frame_pipelining_changes_timing_not_results(weights_dir):
    """sship_sp_ring_submit (frame t+1's extraction enqueued before frame t's match): the same matches per frame as the strictly
    sequential run, with extractions really submitted ahead."""
    runs = {}
    for flag in ((), ("--no-pipeline",)):
        out = subprocess.run([_build(), "--sp", weights_dir["sp_path"], "--lg", weights_dir["lg_path"], "--synthetic", "40",
                              "--keyframe-match", *flag], capture_output=True, text=True, timeout=300)
        print(out.stdout)
        assert out.returncode == 0, out.stdout + out.stderr
        m = re.search(r"stereo matches\s*:\s*([0-9.]+) per frame, ([0-9.]+) pass", out.stdout)
        k = re.search(r"keyframe matches\s*:\s*([0-9.]+)", out.stdout)
        a = re.search(r"\((\d+) of (\d+) extractions enqueued one frame ahead\)", out.stdout)
        runs[bool(flag)] = (m.group(1), m.group(2), k.group(1), int(a.group(1)), _field(out.stdout, "per-frame ms\\s+mean="))
    piped, seq = runs[False], runs[True]
    assert piped[:3] == seq[:3], (piped, seq)       # identical stereo / gated / keyframe match counts
    assert seq[3] == 0 and piped[3] >= 20           # most frames were enqueued ahead (the decoder keeps up with a memcpy)
    print(f"per-frame ms: pipelined {piped[4]:.3f}, sequential {seq[4]:.3f}")


@pytest.mark.gpu
def test_benchmark_runner_reads_a_pgm_sequence(weights_dir, tmp_path):
    from superslam_amd.synth import make_stereo_pair

    for cam in ("image_0", "image_1"):
        os.makedirs(tmp_path / cam)
    for i in range(3):
        l, r = make_stereo_pair(200, 328, 100 + i)
        for cam, im in (("image_0", l), ("image_1", r)):
            with open(tmp_path / cam / f"{i:06d}.pgm", "wb") as f:
                f.write(b"P5\n# written by the test\n%d %d\n255\n" % (im.shape[1], im.shape[0]))
                f.write(np.ascontiguousarray(im).tobytes())
    out = subprocess.run([_build(), "--sp", weights_dir["sp_path"], "--lg", weights_dir["lg_path"], "--sequence", str(tmp_path)],
                         capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert _field(out.stdout, "frames") == 3


def write_png(path, img, filter_type=None, color=False, sixteen=False):
    """Minimal PNG writer (zlib + the five row filters) for the decoder tests: img u8 [H,W] (gray) or [H,W,3] (RGB)."""
    import struct
    import zlib

    a = np.ascontiguousarray(img)
    if sixteen:
        a = np.stack([a, np.zeros_like(a)], -1)          # big-endian 16-bit samples: high byte = the 8-bit value
    h, w = a.shape[:2]
    row = a.reshape(h, -1).astype(np.int32)
    bpp = row.shape[1] // w
    raw = bytearray()
    prev = np.zeros(row.shape[1], np.int32)
    for y in range(h):
        ft = (y % 5) if filter_type is None else filter_type
        cur = row[y]
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
        ul = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
        if ft == 0:
            out = cur
        elif ft == 1:
            out = cur - left
        elif ft == 2:
            out = cur - prev
        elif ft == 3:
            out = cur - ((left + prev) >> 1)
        else:
            p = left + prev - ul
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - ul)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            out = cur - pred
        raw.append(ft)
        raw += (out & 0xff).astype(np.uint8).tobytes()
        prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xffffffff)

    ihdr = struct.pack(">IIBBBBB", w, h, 16 if sixteen else 8, 2 if color else 0, 0, 0, 0)
    comp = zlib.compress(bytes(raw), 6)
    with open(path, "wb") as f:   # two IDAT chunks: the decoder must concatenate them
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", ihdr) + chunk(b"IDAT", comp[: len(comp) // 2]) + chunk(b"IDAT", comp[len(comp) // 2:])
                + chunk(b"IEND", b""))


def _build_io_test(sanitize=False):
    from _cppbuild import cpp_binary

    return cpp_binary("test_image_io", [os.path.join(ROOT, "tests", "cpp", "test_image_io.cc")],
                      deps=[os.path.join(ROOT, "include", "superslam_hip", "image_io.hpp")], link_lib=False, extra=["-lz"], sanitize=sanitize)


def test_png_and_pgm_decoders_bit_exact(tmp_path):
    """include/superslam_hip/image_io.hpp against PNGs written here with every row filter, split IDAT, gray / RGB / 16-bit."""
    from oracle import hostpath as H
    from superslam_amd.synth import make_frame

    g = make_frame(37, 53, 5, n_rects=6)
    rgb = np.stack([make_frame(37, 53, 6, n_rects=6), g, make_frame(37, 53, 7, n_rects=6)], -1)
    cases = {}
    for ft in (None, 0, 1, 2, 3, 4):
        name = f"gray_f{ft}.png"
        write_png(tmp_path / name, g, ft)
        cases[name] = g
    write_png(tmp_path / "rgb.png", rgb, None, color=True)
    cases["rgb.png"] = H.bgr2gray_u8(rgb[..., ::-1])      # cv::imread gives BGR, track_stereo converts BGR2GRAY
    write_png(tmp_path / "gray16.png", g, None, sixteen=True)
    cases["gray16.png"] = g
    with open(tmp_path / "plain.pgm", "wb") as f:
        f.write(b"P5\n# comment\n53 37\n255\n" + g.tobytes())
    cases["plain.pgm"] = g
    with open(tmp_path / "times.txt", "w") as f:
        f.write("0.000000e+00\n1.036224e-01\n2.070932e-01\n\n9.9\n")
    for name, want in cases.items():
        out = subprocess.run([_build_io_test(), str(tmp_path / name)], capture_output=True, timeout=60)
        assert out.returncode == 0, (name, out.stderr)
        hdr, _, body = out.stdout.partition(b"\n")
        r, c = map(int, hdr.split())
        got = np.frombuffer(body, np.uint8).reshape(r, c)
        np.testing.assert_array_equal(got, want, err_msg=name)
    out = subprocess.run([_build_io_test(), "--times", str(tmp_path / "times.txt")], capture_output=True, text=True, timeout=60)
    assert out.stdout.split() == ["3", "0.000000", "0.103622", "0.207093"]     # stops at the first empty line, as the reference
    bad = tmp_path / "bad.png"
    bad.write_bytes(b"not a png")
    assert subprocess.run([_build_io_test(), str(bad)], capture_output=True, timeout=60).returncode == 1


def test_png_decoder_reads_files_of_an_independent_encoder(tmp_path):
    """The same decoder against PNGs written by Pillow (adaptive row filters, several zlib levels, `optimize`, a KITTI-sized
    1241 x 376 frame, RGB, RGBA, 16-bit gray) - an encoder this repository's author did not write.  Modes the runner does not
    support (palette, Adam7 interlace) must be refused with exit status 1, not decoded into garbage."""
    PIL = pytest.importorskip("PIL.Image")
    from oracle import hostpath as H
    from superslam_amd.synth import make_frame

    g = make_frame(376, 1241, 8, n_rects=40)
    small = make_frame(61, 47, 9, n_rects=5)
    rgb = np.stack([make_frame(61, 47, 10, n_rects=5), small, make_frame(61, 47, 11, n_rects=5)], -1)
    cases = {}
    for lvl, opt in ((1, False), (6, False), (9, True)):
        name = f"kitti_l{lvl}.png"
        PIL.fromarray(g, "L").save(tmp_path / name, compress_level=lvl, optimize=opt)
        cases[name] = g
    PIL.fromarray(rgb, "RGB").save(tmp_path / "rgb.png")
    cases["rgb.png"] = H.bgr2gray_u8(rgb[..., ::-1])
    rgba = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 200, np.uint8)], -1)
    PIL.fromarray(rgba, "RGBA").save(tmp_path / "rgba.png")
    cases["rgba.png"] = H.bgr2gray_u8(rgb[..., ::-1])           # cv::imread(IMREAD_COLOR) drops alpha
    PIL.fromarray(small.astype(np.uint16) * 257).save(tmp_path / "g16.png")  # uint16 -> mode I;16
    cases["g16.png"] = small                                     # cv::imread without IMREAD_ANYDEPTH scales 16 -> 8 bit
    for name, want in cases.items():
        out = subprocess.run([_build_io_test(), str(tmp_path / name)], capture_output=True, timeout=60)
        assert out.returncode == 0, (name, out.stderr)
        hdr, _, body = out.stdout.partition(b"\n")
        r, c = map(int, hdr.split())
        np.testing.assert_array_equal(np.frombuffer(body, np.uint8).reshape(r, c), want, err_msg=name)
    PIL.fromarray(small, "L").convert("P").save(tmp_path / "palette.png")
    out = subprocess.run([_build_io_test(), str(tmp_path / "palette.png")], capture_output=True, timeout=60)
    assert out.returncode == 1, out.stdout[:64]


@pytest.mark.gpu
def test_benchmark_runner_reads_a_kitti_style_png_sequence_through_the_upload_ring(weights_dir, tmp_path):
    from superslam_amd.synth import make_stereo_pair

    for cam in ("image_0", "image_1"):
        os.makedirs(tmp_path / cam)
    with open(tmp_path / "times.txt", "w") as f:
        for i in range(6):
            f.write("%e\n" % (0.1 * i))
    for i in range(7):    # one more frame on disk than times.txt announces
        l, r = make_stereo_pair(200, 328, 100 + i)
        write_png(tmp_path / "image_0" / f"{i:06d}.png", l)
        write_png(tmp_path / "image_1" / f"{i:06d}.png", r)
    res = {}
    for mode in ([], ["--no-ring"]):
        out = subprocess.run([_build(), "--sp", weights_dir["sp_path"], "--lg", weights_dir["lg_path"], "--sequence", str(tmp_path), *mode],
                             capture_output=True, text=True, timeout=300)
        print(out.stdout)
        assert out.returncode == 0, out.stdout + out.stderr
        assert _field(out.stdout, "frames") == 6 and "PNG sequence" in out.stdout and "times.txt" in out.stdout
        assert ("pinned upload ring" in out.stdout) == (not mode)
        m = re.search(r"stereo matches\s*:\s*([0-9.]+) per frame, ([0-9.]+) pass", out.stdout)
        res[bool(mode)] = (float(m.group(1)), float(m.group(2)))
    assert res[False] == res[True]      # the ring path and the copying path see the same pixels
