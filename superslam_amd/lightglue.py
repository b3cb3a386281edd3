"""LightGlue matcher: host-side mirror of the reference class (include/LightGlue.h:28-63).

  LightGlueEngine                      - shareable weights (the deserialized-engine analogue)
  LightGlue(engine_file | engine, w, h) - initialize(), shared_engine(), match(...), descriptors_to_host(...)
match() accepts host descriptors (float32 [N,256] - the loop-closure overload, src/LightGlue.cc:285-324)
or DeviceDescriptors (live tracking, :377-457) and returns a MatchResult whose matches are
(queryIdx, trainIdx, distance = 1 - score) rows in ascending queryIdx (:326-363).
Interface methods never raise on runtime failures: they return an empty MatchResult (:381-391).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .pool import DeviceDescriptors


@dataclass
class MatchResult:
    """::MatchResult (include/InferenceInterfaces.h:12-15); `scores` is never filled by the reference."""
    query_idx: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    train_idx: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    distance: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))
    matches0: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))    # raw engine outputs
    mscores0: np.ndarray = field(default_factory=lambda: np.zeros(0, np.float32))

    def __len__(self):
        return len(self.query_idx)


class LightGlueEngine:
    def __init__(self, engine_file: str):
        if not _lib._inited:
            _lib.init()
        self._h = C.c_void_p()
        _lib.check(_lib.lib().sship_lg_weights_load(engine_file.encode(), C.byref(self._h)))

    def __del__(self):
        try:
            if self._h:
                _lib.lib().sship_lg_weights_release(self._h)
        except Exception:
            pass


class LightGlue:
    def __init__(self, engine, image_width: int, image_height: int, max_keypoints: int = 1024, max_pairs: int = 1):
        self._engine_arg = engine
        self.image_width, self.image_height = int(image_width), int(image_height)
        self.max_keypoints, self.max_pairs = int(max_keypoints), int(max_pairs)
        self._engine = engine if isinstance(engine, LightGlueEngine) else None
        self._h = None
        self.last_error = ""

    def initialize(self) -> bool:
        try:
            if self._engine is None:
                self._engine = LightGlueEngine(self._engine_arg)
            h = C.c_void_p()
            _lib.check(_lib.lib().sship_lg_create(self._engine._h, self.image_width, self.image_height,
                                                  self.max_keypoints, self.max_pairs, C.byref(h)))
            self._h = h
            return True
        except _lib.SshipError as e:
            self.last_error = str(e)
            return False

    def shared_engine(self) -> LightGlueEngine:
        return self._engine

    def close(self):
        if self._h is not None:
            _lib.lib().sship_lg_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def normalize_keypoints(self, kp: np.ndarray) -> np.ndarray:
        k = np.ascontiguousarray(kp, np.float32)
        out = np.zeros((k.shape[0], 2), np.float32)
        _lib.check(_lib.lib().sship_lg_normalize_keypoints(self._h, k.ctypes.data, k.shape[1], k.shape[0],
                                                           out.ctypes.data))
        return out

    def match(self, kp0: np.ndarray, d0, kp1: np.ndarray, d1) -> MatchResult:
        r = MatchResult()
        if self._h is None:
            return r
        n0, n1 = len(kp0), len(kp1)
        if n0 == 0 or n1 == 0:
            return r
        k0 = np.ascontiguousarray(kp0, np.float32).reshape(n0, -1)
        k1 = np.ascontiguousarray(kp1, np.float32).reshape(n1, -1)
        m0 = np.full(n0, -1, np.int32)
        ms0 = np.zeros(n0, np.float32)
        L = _lib.lib()
        if isinstance(d0, DeviceDescriptors):
            if d0.empty() or d1.empty():
                return r
            rc = L.sship_lg_match_device(self._h, k0.ctypes.data, k0.shape[1], n0, d0.data, k1.ctypes.data,
                                         k1.shape[1], n1, d1.data, m0.ctypes.data, ms0.ctypes.data)
        else:
            a0 = np.ascontiguousarray(d0, np.float32)
            a1 = np.ascontiguousarray(d1, np.float32)
            rc = L.sship_lg_match_host(self._h, k0.ctypes.data, k0.shape[1], n0, a0.ctypes.data, k1.ctypes.data,
                                       k1.shape[1], n1, a1.ctypes.data, m0.ctypes.data, ms0.ctypes.data)
        if rc != _lib.OK:
            self.last_error = (L.sship_last_error() or b"").decode()
            return r
        q = np.zeros(n0, np.int32)
        t = np.zeros(n0, np.int32)
        d = np.zeros(n0, np.float32)
        k = L.sship_filter_matches(m0.ctypes.data, ms0.ctypes.data, n0, q.ctypes.data, t.ctypes.data, d.ctypes.data)
        return MatchResult(q[:k], t[:k], d[:k], m0, ms0)

    def descriptors_to_host(self, d: DeviceDescriptors) -> np.ndarray:
        """src/LightGlue.cc:460-475: fp16 slot -> float32 [count, dim]; empty handle -> empty array."""
        if d.empty():
            return np.zeros((0, 0), np.float32)
        out = np.zeros((d.count, d.dim), np.float32)
        _lib.check(_lib.lib().sship_desc_to_host(d.data, d.count, d.dim, out.ctypes.data))
        return out

    # ---- test-only introspection (include/sship.h sship_lg_debug_*): the parity suite compares internals with the oracle
    DEBUG_X, DEBUG_SIM, DEBUG_KPTS, DEBUG_ROPE = 0, 1, 2, 3

    def debug_set_layers(self, n_layers: int) -> None:
        _lib.check(_lib.lib().sship_lg_debug_set_layers(self._h, int(n_layers)))

    def debug_read(self, what: int, index: int, rows: int, cols: int) -> np.ndarray:
        out = np.zeros((rows, cols), np.float32)
        _lib.check(_lib.lib().sship_lg_debug_read(self._h, what, index, rows, cols, out.ctypes.data))
        return out

    def match_batch_device(self, kp, n, desc, matches0=None, mscores0=None, stream=None):
        """kp f32 [2P,K,3], n i32 [2P], desc f16 [2P,K,256] (torch CUDA) -> matches0 i32 [P,K], mscores0 f32 [P,K]."""
        import torch

        pairs = kp.shape[0] // 2
        k = self.max_keypoints
        if matches0 is None:
            matches0 = torch.empty((pairs, k), dtype=torch.int32, device=kp.device)
        if mscores0 is None:
            mscores0 = torch.empty((pairs, k), dtype=torch.float32, device=kp.device)
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().sship_lg_match_batch_device(self._h, kp.data_ptr(), n.data_ptr(), desc.data_ptr(),
                                                          pairs, matches0.data_ptr(), mscores0.data_ptr(), s))
        return matches0, mscores0
