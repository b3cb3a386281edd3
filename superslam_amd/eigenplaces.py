"""EigenPlaces place recogniser: host-side mirror of the reference class (include/EigenPlaces.h:19-40) and of
superslam::CosineDescriptorIndex / TemporalConsistencyVoter (include/PlaceRecognizer.h, src/PlaceRecognizer.cc).

  EigenPlaces(engine_file, input_width, input_height); initialize() -> bool;
  compute_global_descriptor(image) -> float32 [512] (empty array when not initialised);
  add(keyframe_id, descriptor); query(descriptor, exclude_recent, top_k) -> [(keyframe_id, score)] by descending score.
`engine_file` is the safetensors state dict utils/convert_eigenplaces_to_onnx.py:99 saves (the .engine's replacement).
Preprocessing (src/EigenPlaces.cc:123-145) runs on the host, as in the reference, through the library's own implementation
(sship_ep_preprocess); the network runs on the GPU (sship_ep_infer)."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib


def preprocess(image: np.ndarray, input_w: int, input_h: int) -> np.ndarray:
    img = np.ascontiguousarray(image, np.uint8)
    if img.ndim == 2:
        h, w, ch = img.shape[0], img.shape[1], 1
    else:
        h, w, ch = img.shape
    out = np.zeros((3, input_h, input_w), np.float32)
    _lib.check(_lib.lib().sship_ep_preprocess(img.ctypes.data, h, w, w * ch, ch, input_w, input_h, out.ctypes.data))
    return out


class CosineDescriptorIndex:
    def __init__(self):
        self._ids, self._db = [], []

    @staticmethod
    def _row(d):
        r = np.asarray(d, np.float32).reshape(-1)
        n = float(np.sqrt((r.astype(np.float64) ** 2).sum()))
        return r / np.float32(n) if n > 1e-12 else r.copy()

    def add(self, keyframe_id: int, global_descriptor) -> None:
        self._ids.append(int(keyframe_id))
        self._db.append(self._row(global_descriptor))

    def query(self, global_descriptor, exclude_recent: int, top_k: int, min_score: float):
        m = len(self._ids)
        if m == 0 or m <= exclude_recent:
            return []
        q = self._row(global_descriptor)
        limit = m - exclude_recent
        scores = np.stack(self._db[:limit]) @ q
        out = [(self._ids[i], float(scores[i])) for i in range(limit) if scores[i] >= min_score]
        out.sort(key=lambda t: -t[1])
        return out[:top_k] if top_k > 0 else out

    def size(self) -> int:
        return len(self._ids)


class TemporalConsistencyVoter:
    def __init__(self, required_votes: int, id_tolerance: int):
        self._required, self._tol, self._streak, self._last, self._have = required_votes, id_tolerance, 0, 0, False

    def vote(self, best) -> bool:
        """best: (keyframe_id, score) or None."""
        if best is None:
            self._streak, self._have = 0, False
            return False
        kid = best[0]
        consistent = self._have and abs(kid - self._last) <= self._tol
        self._streak = self._streak + 1 if consistent else 1
        self._last, self._have = kid, True
        return self._streak >= self._required


class EigenPlaces:
    def __init__(self, engine_file: str, input_width: int, input_height: int):
        self.engine_file, self.input_width, self.input_height = engine_file, int(input_width), int(input_height)
        self.min_score = float(os.environ.get("SUPERSLAM_LOOP_MIN_SCORE", 0.75))   # src/EigenPlaces.cc:33-34
        self._h = None
        self._index = CosineDescriptorIndex()
        self.last_error = ""

    def initialize(self) -> bool:
        try:
            if not _lib._inited:
                _lib.init()
            h = C.c_void_p()
            _lib.check(_lib.lib().sship_ep_create(self.engine_file.encode(), self.input_width, self.input_height, C.byref(h)))
            self._h = h
            return True
        except _lib.SshipError as e:
            self.last_error = str(e)
            return False

    def close(self):
        if self._h is not None:
            _lib.lib().sship_ep_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def compute_global_descriptor(self, image: np.ndarray) -> np.ndarray:
        if self._h is None:
            return np.zeros(0, np.float32)
        x = preprocess(image, self.input_width, self.input_height)
        d = np.zeros(512, np.float32)
        rc = _lib.lib().sship_ep_infer(self._h, x.ctypes.data, d.ctypes.data)
        if rc != _lib.OK:
            self.last_error = (_lib.lib().sship_last_error() or b"").decode()
            return np.zeros(0, np.float32)
        n = float(np.linalg.norm(d))
        return d / np.float32(n) if n > 0 else d     # cv::normalize(desc, desc, 1.0, 0.0, NORM_L2)

    def add(self, keyframe_id: int, global_descriptor) -> None:
        self._index.add(keyframe_id, global_descriptor)

    def query(self, global_descriptor, exclude_recent: int, top_k: int):
        return self._index.query(global_descriptor, exclude_recent, top_k, self.min_score)
