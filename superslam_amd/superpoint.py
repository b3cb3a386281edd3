"""SuperPoint extractor: host-side mirror of the reference class (include/SuperPoint.h:36-54).

Same constructor arguments, method names and error behaviour as the reference's C++ class, with
numpy arrays standing in for cv::Mat / std::vector<cv::KeyPoint>:
  SuperPoint(engine_file, max_keypoints, keypoint_threshold, remove_borders); initialize() -> bool;
  infer(image) -> (ok, keypoints [N,3], descriptors f32 [N,256]);   extract(image) -> Features;
  extract_stereo(left, right) -> (Features, Features).
`engine_file` is the path of a safetensors weight file (the TensorRT .engine's replacement).
Interface methods never raise: failures log and return empty results (src/SuperPoint.cc:895-899).
The torch tensors handled here are only containers for device memory.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib
from .pool import DeviceDescriptors


@dataclass
class Features:
    """superslam::Features (include/InferenceInterfaces.h:21-24); keypoints rows are (x, y, response)."""
    keypoints: np.ndarray = field(default_factory=lambda: np.zeros((0, 3), np.float32))
    descriptors: DeviceDescriptors = field(default_factory=DeviceDescriptors)


class SuperPoint:
    descriptor_dim = 256

    def __init__(self, engine_file: str, max_keypoints: int, keypoint_threshold: float, remove_borders: int,
                 nms_radius: int = 4, pool_slots: int = 8, max_batch: int = 2):
        self.engine_file = engine_file
        self.max_keypoints = int(max_keypoints)
        self.keypoint_threshold = float(keypoint_threshold)
        self.remove_borders = int(remove_borders)
        self.nms_radius = int(nms_radius)
        self.pool_slots = pool_slots
        self.max_batch = max_batch
        self._h = None
        self.last_error = ""

    # ---- lifecycle -------------------------------------------------------------------------
    def initialize(self) -> bool:
        try:
            if not _lib._inited:
                _lib.init()
            cfg = _lib.SpConfig(self.engine_file.encode(), self.max_keypoints, self.keypoint_threshold,
                                self.remove_borders, self.nms_radius, self.pool_slots, self.max_batch)
            h = C.c_void_p()
            _lib.check(_lib.lib().sship_sp_create(C.byref(cfg), C.byref(h)))
            self._h = h
            return True
        except _lib.SshipError as e:
            self.last_error = str(e)
            return False

    def close(self):
        if self._h is not None:
            _lib.lib().sship_sp_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def pool_handle(self):
        return _lib.lib().sship_sp_pool(self._h)

    def pool_in_use(self) -> int:
        return _lib.lib().sship_pool_in_use(self.pool_handle)

    # ---- reference interface ---------------------------------------------------------------
    @staticmethod
    def _img_args(image: np.ndarray):
        img = np.ascontiguousarray(image, np.uint8)
        if img.ndim == 2:
            h, w, ch = img.shape[0], img.shape[1], 1
        elif img.ndim == 3 and img.shape[2] in (1, 3):
            h, w, ch = img.shape
        else:
            raise ValueError("image must be [H,W] or [H,W,3] uint8")
        return img, h, w, ch

    def _wrap(self, f: _lib.Features, kp: np.ndarray) -> Features:
        d = DeviceDescriptors(data=f.desc_dev or 0, count=f.n, dim=256, slot=f.slot,
                              pool=self.pool_handle if f.slot >= 0 else None)
        return Features(keypoints=kp[: f.n].copy(), descriptors=d)

    def extract(self, image: np.ndarray) -> Features:
        """SuperPoint::extract (src/SuperPoint.cc:895-899)."""
        if self._h is None:
            return Features()
        img, h, w, ch = self._img_args(image)
        kp = np.zeros((self.max_keypoints, 3), np.float32)
        f = _lib.Features(kp.ctypes.data_as(C.POINTER(C.c_float)), 0, None, -1)
        rc = _lib.lib().sship_sp_extract(self._h, img.ctypes.data, h, w, w * ch, ch, C.byref(f))
        if rc != _lib.OK:
            self.last_error = (_lib.lib().sship_last_error() or b"").decode()
        return self._wrap(f, kp)

    def extract_stereo(self, left: np.ndarray, right: np.ndarray):
        """SuperPoint::extract_stereo (src/SuperPoint.cc:902-908): one batch-2 pass."""
        if self._h is None:
            return Features(), Features()
        l, h, w, ch = self._img_args(left)
        r, h2, w2, ch2 = self._img_args(right)
        if (h, w, ch) != (h2, w2, ch2):
            self.last_error = "SuperPoint: stereo pair must share resolution (rectified)"  # SuperPoint.cc:762-765
            return Features(), Features()
        kl = np.zeros((self.max_keypoints, 3), np.float32)
        kr = np.zeros((self.max_keypoints, 3), np.float32)
        fl = _lib.Features(kl.ctypes.data_as(C.POINTER(C.c_float)), 0, None, -1)
        fr = _lib.Features(kr.ctypes.data_as(C.POINTER(C.c_float)), 0, None, -1)
        rc = _lib.lib().sship_sp_extract_stereo(self._h, l.ctypes.data, r.ctypes.data, h, w, w * ch, ch,
                                                C.byref(fl), C.byref(fr))
        if rc != _lib.OK:
            self.last_error = (_lib.lib().sship_last_error() or b"").decode()
        return self._wrap(fl, kl), self._wrap(fr, kr)

    def infer(self, image: np.ndarray):
        """SuperPoint::infer host path (src/SuperPoint.cc:322-348): (ok, keypoints [N,3], desc f32 [N,256])."""
        if self._h is None:
            return False, np.zeros((0, 3), np.float32), np.zeros((0, 256), np.float32)
        img, h, w, ch = self._img_args(image)
        kp = np.zeros((self.max_keypoints, 3), np.float32)
        desc = np.zeros((self.max_keypoints, 256), np.float32)
        n = C.c_int(0)
        rc = _lib.lib().sship_sp_infer_host(self._h, img.ctypes.data, h, w, w * ch, ch, kp.ctypes.data,
                                            desc.ctypes.data, C.byref(n))
        if rc != _lib.OK:
            self.last_error = (_lib.lib().sship_last_error() or b"").decode()
            return False, kp[:0], desc[:0]
        return True, kp[: n.value].copy(), desc[: n.value].copy()

    # ---- decode-ahead upload ring + cross-frame pipelining (include/sship.h: sship_sp_ring_*) ------------------
    def ring_create(self, depth: int, h: int, w: int, channels: int = 1) -> bool:
        rc = _lib.lib().sship_sp_ring_create(self._h, depth, h, w, channels)
        if rc != _lib.OK:
            self.last_error = (_lib.lib().sship_last_error() or b"").decode()
            self._ring = None   # a failed create leaves no ring: ring_host must not dereference the NULL it would get
            return False
        self._ring = (depth, h, w, channels)
        return True

    def ring_host(self, slot: int, image: int) -> np.ndarray:
        """The slot's pinned host image (0 = left, 1 = right) as a writable numpy view [h, w(, 3)]."""
        if getattr(self, "_ring", None) is None:
            raise RuntimeError("ring_host: no upload ring (ring_create was not called or failed)")
        _, h, w, ch = self._ring
        ptr = _lib.lib().sship_sp_ring_host(self._h, slot, image)
        if not ptr:
            raise IndexError(f"ring_host: no such slot / image ({slot}, {image})")
        buf = (C.c_uint8 * (h * w * ch)).from_address(ptr)
        a = np.frombuffer(buf, np.uint8)
        return a.reshape(h, w) if ch == 1 else a.reshape(h, w, ch)

    def ring_upload(self, slot: int) -> None:
        _lib.check(_lib.lib().sship_sp_ring_upload(self._h, slot))

    def ring_submit(self, slot: int) -> None:
        """Enqueue the extraction of an uploaded slot ahead of time; extract_stereo_ring(slot) then only waits for it."""
        _lib.check(_lib.lib().sship_sp_ring_submit(self._h, slot))

    def extract_stereo_ring(self, slot: int):
        kl = np.zeros((self.max_keypoints, 3), np.float32)
        kr = np.zeros((self.max_keypoints, 3), np.float32)
        fl = _lib.Features(kl.ctypes.data_as(C.POINTER(C.c_float)), 0, None, -1)
        fr = _lib.Features(kr.ctypes.data_as(C.POINTER(C.c_float)), 0, None, -1)
        rc = _lib.lib().sship_sp_extract_stereo_ring(self._h, slot, C.byref(fl), C.byref(fr))
        if rc != _lib.OK:
            # the ring is this library's own API (no reference override whose "never throws" contract would apply):
            # a failed collection (pool exhausted, device error) raises instead of handing out half-filled features
            self.last_error = (_lib.lib().sship_last_error() or b"").decode()
            for f in (fl, fr):
                if f.slot >= 0:
                    _lib.lib().sship_pool_release(_lib.lib().sship_sp_pool(self._h), f.slot)
            raise RuntimeError("extract_stereo_ring: " + self.last_error)
        return self._wrap(fl, kl), self._wrap(fr, kr)

    # ---- device-resident batch path (throughput) -----------------------------------------------
    def extract_batch_device(self, imgs, desc_out=None, kp_out=None, n_out=None, stream=None):
        """imgs: torch uint8 CUDA tensor [B,H,W].  Returns (desc f16 [B,K,256], kp f32 [B,K,3], n i32 [B])."""
        import torch

        b, h, w = imgs.shape
        k = self.max_keypoints
        if desc_out is None:
            desc_out = torch.empty((b, k, 256), dtype=torch.float16, device=imgs.device)
        if kp_out is None:
            kp_out = torch.empty((b, k, 3), dtype=torch.float32, device=imgs.device)
        if n_out is None:
            n_out = torch.empty((b,), dtype=torch.int32, device=imgs.device)
        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().sship_sp_extract_batch_device(self._h, imgs.data_ptr(), b, h, w, desc_out.data_ptr(),
                                                            kp_out.data_ptr(), n_out.data_ptr(), s))
        return desc_out, kp_out, n_out

    def dense(self, imgs, want_logits: bool = False):
        """Dense engine outputs: scores f32 [B,8Hc,8Wc] (post-NMS), descriptors f16 [B,256,Hc,Wc] (+ logits)."""
        import torch

        b, h, w = imgs.shape
        hc, wc = h // 8, w // 8
        scores = torch.empty((b, hc * 8, wc * 8), dtype=torch.float32, device=imgs.device)
        desc = torch.empty((b, 256, hc, wc), dtype=torch.float16, device=imgs.device)
        logits = torch.empty((b, 65, hc, wc), dtype=torch.float32, device=imgs.device) if want_logits else None
        s = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().sship_sp_dense(self._h, imgs.data_ptr(), b, h, w, scores.data_ptr(), desc.data_ptr(),
                                             logits.data_ptr() if want_logits else None, s))
        return (scores, desc, logits) if want_logits else (scores, desc)
