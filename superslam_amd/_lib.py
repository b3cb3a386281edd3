"""ctypes binding of libsuperslam_hip.so (the C ABI in include/sship.h).

The product path is the HIP library and nothing else: if it is missing, or no gfx950 device is
visible, every call raises - there is no CPU or PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# The ONE library this package loads.  No environment variable moves it (include/sship.h, "Environment": the product reads
# SUPERSLAM_HIP_DEVICE and SSHIP_RCCL_LIBRARY, nothing else); a developer A/B build of the same ABI is selected explicitly, in code, with
# set_library_path() before the first call (tests/test_gpu_alt_paths.py and scripts/dev/* do that through scripts/_devlib.py).
LIB_PATH = os.path.join(_HERE, "lib", "libsuperslam_hip.so")


def set_library_path(path) -> None:
    """Load `path` instead of the shipped library (developer A/B builds: build.build_variant / build_dev).  None = keep the shipped one.
    Must come before the first call into the library: one process, one library."""
    global LIB_PATH
    if not path:
        return
    path = os.path.abspath(path)
    if _lib is not None and path != LIB_PATH:
        raise SshipError(ERR_INVALID, f"set_library_path({path}): {LIB_PATH} is already loaded in this process")
    LIB_PATH = path

OK = 0
ERR_INVALID, ERR_HIP, ERR_IO, ERR_NOMEM, ERR_POOL_EXHAUSTED, ERR_NO_DEVICE = 1, 2, 3, 4, 5, 6


class SshipError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"sship error {code}: {msg}")
        self.code = code


class SpConfig(C.Structure):
    _fields_ = [("weights_path", C.c_char_p), ("max_keypoints", C.c_int), ("keypoint_threshold", C.c_double),
                ("remove_borders", C.c_int), ("nms_radius", C.c_int), ("pool_slots", C.c_int),
                ("max_batch", C.c_int)]


class Features(C.Structure):
    _fields_ = [("kp_xys", C.POINTER(C.c_float)), ("n", C.c_int), ("desc_dev", C.c_void_p), ("slot", C.c_int)]


_lib = None
vp, ip, fp = C.c_void_p, C.c_int, C.c_float
_SIGS = {
    "sship_init": (ip, [ip]),
    "sship_version": (ip, []),
    "sship_last_error": (C.c_char_p, []),
    "sship_device_synchronize": (ip, []),
    "sship_pool_create": (ip, [ip, ip, ip, C.POINTER(vp)]),
    "sship_pool_destroy": (None, [vp]),
    "sship_pool_retain": (None, [vp]),
    "sship_pool_release_ref": (None, [vp]),
    "sship_pool_acquire": (ip, [vp]),
    "sship_pool_release": (None, [vp, ip]),
    "sship_pool_in_use": (ip, [vp]),
    "sship_pool_slot_ptr": (vp, [vp, ip]),
    "sship_gather_normalize": (ip, [vp, ip, ip, ip, vp, vp, ip, vp, vp]),
    "sship_gather_normalize_hwc": (ip, [vp, ip, ip, ip, vp, vp, ip, vp, vp]),
    "sship_nms": (ip, [vp, ip, ip, ip, ip, vp, vp]),
    "sship_select_topk": (ip, [vp, ip, ip, ip, ip, C.c_double, ip, ip, ip, ip, vp, vp, vp, vp, vp, vp]),
    "sship_sp_create": (ip, [C.POINTER(SpConfig), C.POINTER(vp)]),
    "sship_sp_destroy": (None, [vp]),
    "sship_sp_pool": (vp, [vp]),
    "sship_sp_max_keypoints": (ip, [vp]),
    "sship_sp_extract": (ip, [vp, vp, ip, ip, ip, ip, C.POINTER(Features)]),
    "sship_sp_extract_stereo": (ip, [vp, vp, vp, ip, ip, ip, ip, C.POINTER(Features), C.POINTER(Features)]),
    "sship_sp_infer_host": (ip, [vp, vp, ip, ip, ip, ip, vp, vp, C.POINTER(ip)]),
    "sship_sp_ring_create": (ip, [vp, ip, ip, ip, ip]),
    "sship_sp_ring_host": (vp, [vp, ip, ip]),
    "sship_sp_ring_upload": (ip, [vp, ip]),
    "sship_sp_extract_stereo_ring": (ip, [vp, ip, C.POINTER(Features), C.POINTER(Features)]),
    "sship_sp_ring_submit": (ip, [vp, ip]),
    "sship_sp_extract_batch_device": (ip, [vp, vp, ip, ip, ip, vp, vp, vp, vp]),
    "sship_sp_dense": (ip, [vp, vp, ip, ip, ip, vp, vp, vp, vp]),
    "sship_sp_debug_activation": (ip, [vp, ip, vp, C.c_ulonglong]),
    "sship_lg_weights_load": (ip, [C.c_char_p, C.POINTER(vp)]),
    "sship_lg_weights_retain": (None, [vp]),
    "sship_lg_weights_release": (None, [vp]),
    "sship_lg_create": (ip, [vp, ip, ip, ip, ip, C.POINTER(vp)]),
    "sship_lg_destroy": (None, [vp]),
    "sship_lg_normalize_keypoints": (ip, [vp, vp, ip, ip, vp]),
    "sship_lg_match_device": (ip, [vp, vp, ip, ip, vp, vp, ip, ip, vp, vp, vp]),
    "sship_lg_match_host": (ip, [vp, vp, ip, ip, vp, vp, ip, ip, vp, vp, vp]),
    "sship_lg_match_batch_device": (ip, [vp, vp, vp, vp, ip, vp, vp, vp]),
    "sship_lg_debug_set_layers": (ip, [vp, ip]),
    "sship_lg_debug_read": (ip, [vp, ip, ip, ip, ip, vp]),
    "sship_filter_matches": (ip, [vp, vp, ip, vp, vp, vp]),
    "sship_ep_create": (ip, [C.c_char_p, ip, ip, C.POINTER(vp)]),
    "sship_ep_destroy": (None, [vp]),
    "sship_ep_descriptor_dim": (ip, [vp]),
    "sship_ep_infer": (ip, [vp, vp, vp]),
    "sship_ep_preprocess": (ip, [vp, ip, ip, ip, ip, ip, ip, vp]),
    "sship_ep_infer_u8": (ip, [vp, vp, ip, ip, ip, ip, vp]),
    "sship_ep_infer_u8_device": (ip, [vp, vp, ip, ip, ip, ip, vp, vp]),
    "sship_ep_bench": (ip, [vp, vp, ip, ip, ip, ip, ip, C.POINTER(C.c_float)]),
    "sship_desc_to_host": (ip, [vp, ip, ip, vp]),
    "sship_frontend_batch_device": (ip, [vp, vp, vp, ip, ip, ip, vp, vp, vp, vp, vp, vp]),
    "sship_sp_bench_layer": (ip, [vp, ip, ip, ip, ip, ip, C.POINTER(fp), C.POINTER(C.c_double)]),
    "sship_lg_bench_stage": (ip, [vp, ip, ip, C.POINTER(fp)]),
    "sship_mfma_probe": (ip, [ip, C.POINTER(fp)]),
    "sship_set_profiling": (None, [ip]),
    "sship_get_stage_timings": (ip, [C.POINTER(C.c_char_p), C.POINTER(fp), ip]),
    "sship_set_log_callback": (None, [vp]),
    "sship_comm_unique_id": (ip, [vp]),
    "sship_comm_create": (ip, [vp, ip, ip, C.POINTER(vp)]),
    "sship_comm_destroy": (None, [vp]),
    "sship_comm_rank": (ip, [vp]),
    "sship_comm_world": (ip, [vp]),
    "sship_gather_features_rccl": (ip, [vp, vp, vp, vp, ip, ip, vp, vp, vp, vp]),
}


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise SshipError(ERR_IO, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                     "(hipcc --offload-arch=gfx950); there is no fallback path")
        # ONE HIP runtime per process: torch bundles its own libamdhip64 / libhsa-runtime64, and a second
        # copy (the system one this library was linked against) cannot initialise the GPU once the first has.
        # Importing torch first makes the dynamic loader resolve our DT_NEEDED libamdhip64.so.7 to the copy
        # torch already mapped (same SONAME), exactly like a torch.utils.cpp_extension module.
        import torch  # noqa: F401  (device-memory / stream plumbing only)

        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int) -> None:
    if rc != OK:
        raise SshipError(rc, (lib().sship_last_error() or b"").decode(errors="replace"))


_inited = False


def init(device: int = -1) -> None:
    """Select the device (SUPERSLAM_HIP_DEVICE / LOCAL_RANK aware).  Raises SshipError without a GPU."""
    global _inited
    if device < 0 and "SUPERSLAM_HIP_DEVICE" not in os.environ and "LOCAL_RANK" in os.environ:
        device = int(os.environ["LOCAL_RANK"])
    check(lib().sship_init(device))
    _inited = True


def stage_timings() -> dict:
    labels = (C.c_char_p * 32)()
    ms = (fp * 32)()
    n = lib().sship_get_stage_timings(labels, ms, 32)
    return {labels[i].decode(): ms[i] for i in range(n)}
