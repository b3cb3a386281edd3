"""ctypes front for oracle/hostpath_ref.c (TEST INFRASTRUCTURE - see oracle/__init__.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_c.so")
_lib = None


def build() -> str:
    src = os.path.join(_HERE, "hostpath_ref.c")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE])
    return _SO


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_select_topk.restype = C.c_int
        _lib.orc_select_topk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                         C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_void_p]
        _lib.orc_filter_matches.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def nms_maxpool(scores: np.ndarray, radius: int = 4) -> np.ndarray:
    s = np.ascontiguousarray(scores, np.float32)
    out = np.empty_like(s)
    lib().orc_nms_maxpool(_p(s), C.c_int(s.shape[0]), C.c_int(s.shape[1]), C.c_int(radius), _p(out))
    return out


def select_topk(scores: np.ndarray, input_h: int, input_w: int, thr: float, border: int, max_kp: int,
                desc_h: int, desc_w: int):
    """-> dict(kp [N,3] f32 (x,y,score), hw [N,2] i32, cell_h [N], cell_w [N], n_candidates)."""
    s = np.ascontiguousarray(scores, np.float32)
    kp = np.zeros((max_kp, 3), np.float32)
    hw = np.zeros((max_kp, 2), np.int32)
    ch = np.zeros(max_kp, np.int32)
    cw = np.zeros(max_kp, np.int32)
    nc = C.c_int(0)
    n = lib().orc_select_topk(_p(s), s.shape[0], s.shape[1], input_h, input_w, float(thr), border, max_kp,
                              desc_h, desc_w, _p(kp), _p(hw), _p(ch), _p(cw), C.byref(nc))
    return dict(kp=kp[:n], hw=hw[:n], cell_h=ch[:n], cell_w=cw[:n], n_candidates=nc.value)


def gather_normalize(grid_f16_chw: np.ndarray, cell_h: np.ndarray, cell_w: np.ndarray, tree: bool = True):
    """grid: float16 [C,gh,gw] -> float16 [N,C]."""
    g = np.ascontiguousarray(grid_f16_chw).view(np.uint16)
    c, gh, gw = g.shape
    ch = np.ascontiguousarray(cell_h, np.int32)
    cw = np.ascontiguousarray(cell_w, np.int32)
    out = np.zeros((len(ch), c), np.uint16)
    lib().orc_gather_normalize(_p(g), C.c_int(c), C.c_int(gh), C.c_int(gw), _p(ch), _p(cw), C.c_int(len(ch)),
                               _p(out), C.c_int(1 if tree else 0))
    return out.view(np.float16)


def normalize_kpts(kp_xy: np.ndarray, image_w: int, image_h: int) -> np.ndarray:
    k = np.ascontiguousarray(kp_xy, np.float32)
    out = np.zeros((k.shape[0], 2), np.float32)
    lib().orc_normalize_kpts(_p(k), C.c_int(k.shape[1]), C.c_int(k.shape[0]), C.c_int(image_w),
                             C.c_int(image_h), _p(out))
    return out


def filter_matches(matches0: np.ndarray, mscores0: np.ndarray):
    m = np.ascontiguousarray(matches0, np.int32)
    s = np.ascontiguousarray(mscores0, np.float32)
    q = np.zeros(len(m), np.int32)
    t = np.zeros(len(m), np.int32)
    d = np.zeros(len(m), np.float32)
    k = lib().orc_filter_matches(_p(m), _p(s), C.c_int(len(m)), _p(q), _p(t), _p(d))
    return q[:k], t[:k], d[:k]


def half_to_float(h: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(h).view(np.uint16)
    out = np.empty(a.shape, np.float32)
    lib().orc_half_to_float(_p(a), _p(out), C.c_long(a.size))
    return out


def float_to_half(f: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(f, np.float32)
    out = np.empty(a.shape, np.uint16)
    lib().orc_float_to_half(_p(a), _p(out), C.c_long(a.size))
    return out.view(np.float16)


def bgr2gray_u8(img_bgr: np.ndarray) -> np.ndarray:
    """cv::cvtColor(img, gray, cv::COLOR_BGR2GRAY) on CV_8UC3 (call sites src/SuperPoint.cc:388,771).

    OpenCV itself is a third-party dependency absent from this image; its published 8-bit algorithm (imgproc
    color_rgb.simd.hpp, RGB2Gray<uchar>, stable across 3.x / 4.x) is fixed point with a 14-bit shift:
      gray = (B*1868 + G*9617 + R*4899 + (1 << 13)) >> 14      (0.114, 0.587, 0.299 scaled by 2^14 and rounded)
    """
    a = np.ascontiguousarray(img_bgr, np.uint8).astype(np.int64)
    return ((a[..., 0] * 1868 + a[..., 1] * 9617 + a[..., 2] * 4899 + (1 << 13)) >> 14).astype(np.uint8)
