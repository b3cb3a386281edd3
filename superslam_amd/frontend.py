"""Front-end step over the two interfaces - what StereoFrontEnd::process asks per frame
(src/StereoFrontEnd.cc:10-48): extract_stereo + one device match + disparity / row gates."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from . import _lib


@dataclass
class StereoObservation:
    keypoints_left: np.ndarray   # [N,3]
    u_right: np.ndarray          # [N] NaN where no depth
    has_depth: np.ndarray        # [N] uint8


def process_stereo(extractor, matcher, left: np.ndarray, right: np.ndarray, min_disparity: float = 1.0):
    """StereoFrontEnd::process semantics: uL - uR >= min_disparity and |vL - vR| <= 2."""
    L, R = extractor.extract_stereo(left, right)
    n = len(L.keypoints)
    u_right = np.full(n, np.nan, np.float32)
    has_depth = np.zeros(n, np.uint8)
    m = matcher.match(L.keypoints, L.descriptors, R.keypoints, R.descriptors)
    for i, j in zip(m.query_idx, m.train_idx):
        if i < 0 or j < 0 or i >= n or j >= len(R.keypoints):
            continue
        uL, v = L.keypoints[i, 0], L.keypoints[i, 1]
        uR = R.keypoints[j, 0]
        if uL - uR < min_disparity:
            continue
        if abs(v - R.keypoints[j, 1]) > 2.0:
            continue
        u_right[i] = uR
        has_depth[i] = 1
    return StereoObservation(L.keypoints, u_right, has_depth), L, R, m


class FrontEndBatch:
    """Device-resident throughput step: SuperPoint on 2P images + LightGlue on P pairs, no host sync."""

    def __init__(self, sp, lg, pairs: int, h: int, w: int, device="cuda"):
        import torch

        self.sp, self.lg, self.pairs, self.h, self.w = sp, lg, pairs, h, w
        k = sp.max_keypoints
        self.desc = torch.zeros((2 * pairs, k, 256), dtype=torch.float16, device=device)
        self.kp = torch.zeros((2 * pairs, k, 3), dtype=torch.float32, device=device)
        self.n = torch.zeros((2 * pairs,), dtype=torch.int32, device=device)
        self.matches0 = torch.zeros((pairs, k), dtype=torch.int32, device=device)
        self.mscores0 = torch.zeros((pairs, k), dtype=torch.float32, device=device)

    def run(self, imgs, stream=None):
        """imgs: uint8 CUDA [2P,H,W] ordered L0,R0,L1,R1,...  Asynchronous."""
        import torch

        s = stream if stream is not None else torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().sship_frontend_batch_device(
            self.sp._h, self.lg._h, imgs.data_ptr(), self.pairs, self.h, self.w, self.desc.data_ptr(),
            self.kp.data_ptr(), self.n.data_ptr(), self.matches0.data_ptr(), self.mscores0.data_ptr(), s))
        return self
