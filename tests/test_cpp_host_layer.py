"""The C++ host layer above the C ABI (include/superslam_hip/frontend.hpp): compiled with g++ against the
in-tree .so; CPU part always, GPU part under -m gpu."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "superslam_amd", "lib", "test_host_layer")


def _build():
    libdir = os.path.join(ROOT, "superslam_amd", "lib")
    src = os.path.join(ROOT, "tests", "cpp", "test_host_layer.cc")
    hdr = os.path.join(ROOT, "include", "superslam_hip", "frontend.hpp")
    if not os.path.exists(BIN) or os.path.getmtime(BIN) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), src, "-o", BIN,
                               "-L" + libdir, "-lsuperslam_hip", "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib"])
    return BIN


def test_cpp_host_layer_cpu():
    from superslam_amd import _lib

    _lib.lib()
    out = subprocess.run([_build()], capture_output=True, text=True, timeout=120)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr


@pytest.mark.gpu
def test_cpp_host_layer_gpu(weights_dir):
    out = subprocess.run([_build(), weights_dir["sp_path"], weights_dir["lg_path"]], capture_output=True, text=True,
                         timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
