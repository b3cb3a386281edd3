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


def test_rccl_exchange_entry_points_validate_their_arguments_and_need_no_rccl_at_load_time():
    """The multi-GPU exchange (sship_comm_* / sship_gather_features_rccl) binds RCCL at run time: the library has no DT_NEEDED on
    librccl (single-GPU users need no RCCL), and bad arguments are refused before anything touches RCCL or a device."""
    import ctypes as C

    from superslam_amd import _lib

    lib = _lib.lib()
    h = C.c_void_p()
    assert lib.sship_comm_create(None, 0, 1, C.byref(h)) == _lib.ERR_INVALID
    assert lib.sship_comm_create(b"\0" * 128, 3, 2, C.byref(h)) == _lib.ERR_INVALID          # rank outside the world
    assert lib.sship_comm_unique_id(None) == _lib.ERR_INVALID
    assert lib.sship_gather_features_rccl(None, None, None, None, 4, 600, None, None, None, None) == _lib.ERR_INVALID
    assert b"communicator" in lib.sship_last_error()
    assert lib.sship_comm_rank(None) == -1 and lib.sship_comm_world(None) == 0
    lib.sship_comm_destroy(None)
    needed = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "rccl" not in needed.lower() and "torch" not in needed.lower(), needed


def test_shipped_library_reads_only_the_documented_environment():
    """one kernel per layer in the shipped library.  The only environment variable NAMES inside libsuperslam_hip.so
    are the two include/sship.h documents; every A/B switch and the kernels it selected live in lib/variants/dev.so."""
    from superslam_amd import _lib, build

    blob = open(os.path.join(ROOT, "superslam_amd", "lib", "libsuperslam_hip.so"), "rb").read()
    names = set(re.findall(rb"(?:SUPERSLAM_[A-Z0-9_]{3,}|SSHIP_[A-Z0-9_]{3,})", blob))
    names = {n.decode() for n in names if not n.startswith(b"SSHIP_ERR") and not n.startswith(b"SSHIP_HIP_CHECK")}
    assert names == {"SUPERSLAM_HIP_DEVICE", "SSHIP_RCCL_LIBRARY"}, names
    hdr = open(os.path.join(ROOT, "include", "sship.h")).read()
    assert "SUPERSLAM_HIP_DEVICE" in hdr and "SSHIP_RCCL_LIBRARY" in hdr
    for rejected in ("conv_strip.hip", "conv_wino.hip", "lg_ffn16.hip", "lg_attn_res.hip"):
        assert rejected not in build.SOURCES and rejected in build.DEV_SOURCES
    syms = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    for k in ("conv_wino", "k_lg_ffn16", "k_assign_row_lse", "attention_res"):
        assert k not in syms, k


def test_rccl_missing_is_an_error_code_not_a_crash():
    """with no RCCL to bind, sship_comm_unique_id / sship_comm_create with VALID arguments return
    SSHIP_ERR_NO_DEVICE and a message (the old code called dlerror() twice and built a std::string from NULL).  A fresh process:
    the binding is resolved once per process; SSHIP_RCCL_LIBRARY names the one library to try (include/sship.h, Environment)."""
    code = (
        "import ctypes as C\n"
        "from superslam_amd import _lib\n"
        "lib = _lib.lib()\n"
        "buf = C.create_string_buffer(128)\n"
        "rc = lib.sship_comm_unique_id(buf)\n"
        "assert rc == _lib.ERR_NO_DEVICE, rc\n"
        "msg = lib.sship_last_error()\n"
        "assert b'RCCL not found' in msg and b'/nonexistent/librccl.so' in msg, msg\n"
        "h = C.c_void_p()\n"
        "assert lib.sship_comm_create(buf, 0, 1, C.byref(h)) == _lib.ERR_NO_DEVICE\n"
        "print('ok')\n")
    env = dict(os.environ, SSHIP_RCCL_LIBRARY="/nonexistent/librccl.so", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=env)
    assert out.returncode == 0 and "ok" in out.stdout, out.stdout + out.stderr


def test_single_collective_record_packing_round_trips():
    """all_gather_features sends ONE byte record per unit (descriptor rows | keypoint rows | count, padded to 16 B)."""
    import torch

    from superslam_amd.shard import _pack_units, _unpack_units

    g = torch.Generator().manual_seed(3)
    desc = torch.randn((3, 7, 256), generator=g).half(); kp = torch.rand((3, 7, 3), generator=g)
    n = torch.tensor([7, 0, 4], dtype=torch.int32)
    buf, db, kb = _pack_units(desc, kp, n, per=5)
    assert buf.shape == (5, (7 * 512 + 7 * 12 + 4 + 15) // 16 * 16) and buf.dtype == torch.uint8
    d2, k2, n2 = _unpack_units(buf, db, kb, 7, desc.dtype, kp.dtype, n.dtype)
    assert torch.equal(d2[:3], desc) and torch.equal(k2[:3], kp) and torch.equal(n2[:3], n)
    assert int(n2[3:].abs().sum()) == 0 and float(d2[3:].abs().sum()) == 0.0


def test_python_layer_has_no_environment_override_of_the_library():
    """SUPERSLAM_HIP_LIBRARY used to swap the whole .so behind the package's back.  The package now loads ONE path; a
    developer build is selected in code (set_library_path).  A stray variable must change nothing."""
    import subprocess
    import sys

    code = ("import os, sys; sys.path.insert(0, %r); from superslam_amd import _lib; print(_lib.LIB_PATH); "
            "_lib.set_library_path(None); print(_lib.LIB_PATH); _lib.set_library_path('/tmp/other.so'); print(_lib.LIB_PATH)" % ROOT)
    env = dict(os.environ, SUPERSLAM_HIP_LIBRARY="/nonexistent/evil.so", SSHIP_DEV_LIBRARY="/nonexistent/evil2.so")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120).stdout.split()
    shipped = os.path.join(ROOT, "superslam_amd", "lib", "libsuperslam_hip.so")
    assert out == [shipped, shipped, "/tmp/other.so"], out
    import re

    for f in os.listdir(os.path.join(ROOT, "superslam_amd")):
        if f.endswith(".py"):
            src = open(os.path.join(ROOT, "superslam_amd", f)).read()
            names = set(re.findall(r"environ(?:\.get|\.setdefault)?[\[(]\s*[\"']([A-Z_0-9]+)", src))
            assert not ({n for n in names if "LIBRARY" in n} - {"SSHIP_RCCL_LIBRARY"}), (f, names)
