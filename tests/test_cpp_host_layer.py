"""The C++ host layer above the C ABI (include/superslam_hip/frontend.hpp): compiled with g++ against the
in-tree .so; CPU part always, GPU part under -m gpu."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_build", "test_host_layer")   # test artefact: outside the package directory (git-ignored, travels to the GPU box)


def _build(sanitize=False):
    from _cppbuild import cpp_binary

    return cpp_binary("test_host_layer", [os.path.join(ROOT, "tests", "cpp", "test_host_layer.cc")],
                      deps=[os.path.join(ROOT, "include", "superslam_hip", "frontend.hpp"), os.path.join(ROOT, "include", "sship.h")], sanitize=sanitize)


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
