"""CPU: the C-ABI library loads and exports every symbol include/sship.h declares; the product path fails
loudly without a GPU (no CPU fallback) and never imports the oracle."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "sship.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sship_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from superslam_amd import _lib

    lib = _lib.lib()
    syms = _declared_symbols()
    assert len(syms) >= 35
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing
    # and the Python binding table covers all of them
    assert sorted(_lib._SIGS) == syms
    assert lib.sship_version() == 100


def test_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible here")
    from superslam_amd import SuperPoint, _lib

    lib = _lib.lib()
    assert lib.sship_init(-1) == _lib.ERR_NO_DEVICE
    assert b"no CPU path" in lib.sship_last_error()
    sp = SuperPoint("whatever.safetensors", 600, 0.005, 4)
    assert sp.initialize() is False and "no HIP device" in sp.last_error
    with pytest.raises(_lib.SshipError):
        _lib.init()


def test_product_package_never_imports_the_oracle():
    code = "import sys; import superslam_amd, superslam_amd.frontend, superslam_amd.shard; " \
           "assert not [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')], 'oracle imported'"
    subprocess.check_call([sys.executable, "-c", code], cwd=ROOT)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "superslam_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, f


def test_pure_host_helpers_work_without_gpu():
    """sship_filter_matches is pure host code (src/LightGlue.cc:326-363)."""
    import numpy as np

    from superslam_amd import _lib

    m0 = np.array([3, -1, 0, -1, 7, 2], np.int32)
    ms = np.array([0.9, 0.5, 0.25, 0.0, 1.0, 0.125], np.float32)
    q = np.zeros(6, np.int32); t = np.zeros(6, np.int32); d = np.zeros(6, np.float32)
    k = _lib.lib().sship_filter_matches(m0.ctypes.data, ms.ctypes.data, 6, q.ctypes.data, t.ctypes.data, d.ctypes.data)
    assert k == 4 and q[:4].tolist() == [0, 2, 4, 5] and t[:4].tolist() == [3, 0, 7, 2]
    np.testing.assert_allclose(d[:4], [0.1, 0.75, 0.0, 0.875], atol=1e-7)
