"""EigenPlaces place recogniser, descriptor source only: host-side mirror of the reference class (include/EigenPlaces.h:19-30).

  EigenPlaces(engine_file, input_width, input_height); initialize() -> bool;
  compute_global_descriptor(image) -> float32 [512] (empty array when not initialised).
Retrieval (superslam::CosineDescriptorIndex / TemporalConsistencyVoter, src/PlaceRecognizer.cc) is the reference's own GPU-free control
plane and is not restated in this package (the tests' restatement: oracle/eigenplaces_ref.py).  NOTE for users of rounds <= 4: `add` / `query`
of IPlaceRecognizer (include/EigenPlaces.h:30-36) were removed from this class on purpose in round 5 - in a SuperSLAM build the adapter
integration/reference_side/EigenPlaces.h implements them with the reference's own index.
`engine_file` is the safetensors state dict utils/convert_eigenplaces_to_onnx.py:99 saves (the .engine's replacement).
The u8 image is uploaded and preprocessed on the device (sship_ep_infer_u8: fixed-point 8-bit bilinear resize + normalisation, bit-identical
to the host form sship_ep_preprocess of src/EigenPlaces.cc:123-145); `preprocess` below exposes the host form for the tests."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib


def _u8_image(image) -> np.ndarray:
    """The cv::Mat the reference takes is 8-bit, 1 channel or 3 (BGR).  Anything else is refused, never cast: a float image in [0, 1] cast to
    uint8 is all zeros, and a descriptor of a black frame would be a silently wrong answer."""
    img = np.asarray(image)
    if img.dtype != np.uint8:
        raise TypeError(f"EigenPlaces: expected a uint8 image (cv::Mat CV_8U), got {img.dtype}")
    if not (img.ndim == 2 or (img.ndim == 3 and img.shape[2] in (1, 3))):
        raise ValueError(f"EigenPlaces: expected [H, W] gray or [H, W, 3] BGR, got shape {img.shape}")
    if img.shape[0] == 0 or img.shape[1] == 0:
        raise ValueError("EigenPlaces: empty image")
    return np.ascontiguousarray(img)


def preprocess(image: np.ndarray, input_w: int, input_h: int) -> np.ndarray:
    img = _u8_image(image)
    if img.ndim == 2:
        h, w, ch = img.shape[0], img.shape[1], 1
    else:
        h, w, ch = img.shape
    out = np.zeros((3, input_h, input_w), np.float32)
    _lib.check(_lib.lib().sship_ep_preprocess(img.ctypes.data, h, w, w * ch, ch, input_w, input_h, out.ctypes.data))
    return out


class EigenPlaces:
    def __init__(self, engine_file: str, input_width: int, input_height: int):
        self.engine_file, self.input_width, self.input_height = engine_file, int(input_width), int(input_height)
        self._h = None
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
        img = _u8_image(image)   # TypeError / ValueError for anything but 8-bit gray or BGR (API misuse, not a run-time failure)
        h, w = img.shape[0], img.shape[1]
        ch = 1 if img.ndim == 2 else img.shape[2]
        d = np.zeros(512, np.float32)
        rc = _lib.lib().sship_ep_infer_u8(self._h, img.ctypes.data, h, w, w * ch, ch, d.ctypes.data)
        if rc != _lib.OK:
            self.last_error = (_lib.lib().sship_last_error() or b"").decode()
            return np.zeros(0, np.float32)
        n = float(np.linalg.norm(d))
        return d / np.float32(n) if n > 0 else d     # cv::normalize(desc, desc, 1.0, 0.0, NORM_L2)
