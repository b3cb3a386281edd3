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
    libdir = os.path.join(ROOT, "superslam_amd", "lib")
    src = os.path.join(ROOT, "examples", "frontend_benchmark.cc")
    hdr = os.path.join(ROOT, "include", "superslam_hip", "frontend.hpp")
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-I" + os.path.join(ROOT, "include"), src, "-o", BIN,
                               "-L" + libdir, "-lsuperslam_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lpthread"])
    return BIN


def test_benchmark_runner_builds_and_rejects_bad_usage():
    from superslam_amd import _lib

    _lib.lib()   # the .so must exist (built by __graft_entry__.build / superslam_amd.build)
    out = subprocess.run([_build()], capture_output=True, text=True, timeout=60)
    assert out.returncode == 2 and "usage:" in out.stderr


def _field(text, label):
    m = re.search(label + r"\s*:\s*([0-9.]+)", text)
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
