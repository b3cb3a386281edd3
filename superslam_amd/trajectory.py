"""Trajectory files and the ATE / KITTI-segment evaluation around the front-end (BASELINE configs[3]).

The reference writes its estimate with SuperSLAM::save_trajectory (src/SuperSLAM.cc:192-218: KITTI = one camera-to-world
3x4 [Rwc | twc] per line, row-major, 12 values, std::fixed with 9 decimals; TUM = ``timestamp tx ty tz qx qy qz qw``) and
scores it with scripts/benchmarks/_eval_common.py, which leans on the third-party ``evo`` package (absent here):
``ate`` = evo APE on the translation part after a Umeyama SE(3)/Sim(3) alignment (:38-47), ``kitti_segments`` = the official
KITTI odometry metric over 100..800 m sub-sequences (:72-111).  This module restates both in numpy so a run of the
StereoFrontEnd -> VoEstimator loop with the HIP front-end can be written and scored without evo; the KITTI-00 gate itself
(ATE RMSE 1.582 m, README.md:23) needs the dataset, real weights and the GTSAM estimator, none of which exist offline -
tests/test_trajectory.py checks the writer formats and the metrics against closed-form trajectories.
"""
from __future__ import annotations

import numpy as np

KITTI_LENGTHS = (100, 200, 300, 400, 500, 600, 700, 800)   # _eval_common.py:68
KITTI_STEP = 10                                            # :69


def _as_4x4(poses) -> np.ndarray:
    p = np.asarray(poses, np.float64)
    if p.ndim == 2 and p.shape[1] == 12:
        p = p.reshape(-1, 3, 4)
    if p.shape[1:] == (3, 4):
        bottom = np.tile(np.array([[[0.0, 0.0, 0.0, 1.0]]]), (len(p), 1, 1))
        p = np.concatenate([p, bottom], 1)
    assert p.shape[1:] == (4, 4), p.shape
    return p


def save_trajectory_kitti(path: str, poses_twc) -> None:
    """src/SuperSLAM.cc:199-208: ``R00 R01 R02 tx R10 R11 R12 ty R20 R21 R22 tz`` per pose, fixed, 9 decimals."""
    p = _as_4x4(poses_twc)
    with open(path, "w") as f:
        for T in p:
            f.write(" ".join("%.9f" % v for v in T[:3, :4].reshape(-1)) + "\n")


def rotation_to_quaternion_xyzw(R: np.ndarray) -> np.ndarray:
    """Unit quaternion (x, y, z, w) of a rotation matrix, w >= 0 branch-stable (Shepperd)."""
    R = np.asarray(R, np.float64)
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(1.0 + R[i, i] - R[j, j] - R[k, k]) * 2
        q = np.zeros(4)
        q[i] = 0.25 * s
        q[j] = (R[j, i] + R[i, j]) / s
        q[k] = (R[k, i] + R[i, k]) / s
        q[3] = (R[k, j] - R[j, k]) / s
    return q / np.linalg.norm(q)


def save_trajectory_tum(path: str, timestamps, poses_twc) -> None:
    """src/SuperSLAM.cc:209-217: ``timestamp tx ty tz qx qy qz qw`` (Twc), fixed, 9 decimals; missing timestamps = index."""
    p = _as_4x4(poses_twc)
    with open(path, "w") as f:
        for i, T in enumerate(p):
            ts = float(timestamps[i]) if i < len(timestamps) else float(i)
            q = rotation_to_quaternion_xyzw(T[:3, :3])
            f.write(" ".join("%.9f" % v for v in (ts, *T[:3, 3], *q)) + "\n")


def load_kitti_poses(path: str) -> np.ndarray:
    """[N,4,4] camera-to-world poses from a KITTI pose / trajectory file (12 values per line)."""
    return _as_4x4(np.loadtxt(path, ndmin=2))


def umeyama(src_xyz: np.ndarray, dst_xyz: np.ndarray, with_scale: bool = False):
    """Least-squares similarity (R, t, s) with dst ~ s R src + t (Umeyama 1991) - what evo's ``align`` computes."""
    x, y = np.asarray(src_xyz, np.float64), np.asarray(dst_xyz, np.float64)
    mx, my = x.mean(0), y.mean(0)
    xc, yc = x - mx, y - my
    cov = yc.T @ xc / len(x)
    U, D, Vt = np.linalg.svd(cov)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1.0
    R = U @ S @ Vt
    s = float(np.trace(np.diag(D) @ S) / ((xc ** 2).sum() / len(x))) if with_scale else 1.0
    t = my - s * R @ mx
    return R, t, s


def ate(gt_poses, est_poses, align: bool = True, correct_scale: bool = False) -> dict:
    """Absolute trajectory error on the translation part (_eval_common.py:38-47): evo's APE statistics."""
    gt, est = _as_4x4(gt_poses)[:, :3, 3], _as_4x4(est_poses)[:, :3, 3]
    assert len(gt) == len(est) and len(gt) >= 3
    if align:
        R, t, s = umeyama(est, gt, correct_scale)
        est = (s * (R @ est.T)).T + t
    e = np.linalg.norm(gt - est, axis=1)
    return {"rmse": float(np.sqrt((e ** 2).mean())), "mean": float(e.mean()), "median": float(np.median(e)),
            "std": float(e.std()), "min": float(e.min()), "max": float(e.max()), "sse": float((e ** 2).sum())}


def _rigid_inverse(T: np.ndarray) -> np.ndarray:
    """Batched inverse of [..., 4, 4] rigid transforms: [R | t]^-1 = [R^T | -R^T t] (no general matrix inversion)."""
    Rt = np.swapaxes(T[..., :3, :3], -1, -2)
    out = np.zeros_like(T)
    out[..., :3, :3] = Rt
    out[..., :3, 3] = -np.einsum("...ij,...j->...i", Rt, T[..., :3, 3])
    out[..., 3, 3] = 1.0
    return out


def kitti_segment_table(gt_poses, est_poses, lengths=KITTI_LENGTHS, step: int = KITTI_STEP) -> np.ndarray:
    """One row ``(start frame, end frame, segment length [m], translation error [m], rotation error [rad])`` per KITTI
    odometry sub-sequence - the table the KITTI devkit averages (evaluate_odometry: start frames every `step` frames, a
    segment of nominal length L ends at the first frame whose driven distance from the start is >= L; starts whose
    segment runs off the end of the sequence are dropped).

    Formulation: the driven distance s[i] is monotone, so every segment end is one ``np.searchsorted(s, s[start] + L)``
    (all starts x all lengths at once); the segment error E = (est_a^-1 est_b)^-1 (gt_a^-1 gt_b) is evaluated for all
    segments in one batch with the closed-form rigid inverse.  Translation error = |E_t|, rotation error = the angle of E_R.
    """
    gt, est = _as_4x4(gt_poses), _as_4x4(est_poses)
    assert len(gt) == len(est)
    s = np.zeros(len(gt))
    np.cumsum(np.linalg.norm(gt[1:, :3, 3] - gt[:-1, :3, 3], axis=1), out=s[1:])
    starts = np.arange(0, len(gt), step)
    L = np.asarray(lengths, np.float64)
    ends = np.searchsorted(s, s[starts, None] + L[None, :], side="left")      # [starts, lengths]; len(gt) = ran off the end
    # searchsorted looks at the whole sequence; a segment must end at or after its start (s is non-decreasing, L > 0: it does)
    ok = ends < len(gt)
    a = np.broadcast_to(starts[:, None], ends.shape)[ok]
    b = ends[ok]
    seg_len = np.broadcast_to(L[None, :], ends.shape)[ok]
    gt_motion = _rigid_inverse(gt[a]) @ gt[b]
    est_motion = _rigid_inverse(est[a]) @ est[b]
    E = _rigid_inverse(est_motion) @ gt_motion
    t_err = np.linalg.norm(E[:, :3, 3], axis=1)
    # rotation angle from both the trace (cos) and the antisymmetric part (sin): unlike arccos alone this keeps its
    # precision near 0, where the metric of a good estimate lives (arccos(1 - 1e-16) is already 1.5e-8 rad)
    Re = E[:, :3, :3]
    sin_axis = 0.5 * np.stack([Re[:, 2, 1] - Re[:, 1, 2], Re[:, 0, 2] - Re[:, 2, 0], Re[:, 1, 0] - Re[:, 0, 1]], axis=1)
    angle = np.arctan2(np.linalg.norm(sin_axis, axis=1), 0.5 * (np.einsum("nii->n", Re) - 1.0))
    return np.stack([a.astype(np.float64), b.astype(np.float64), seg_len, t_err, angle], axis=1)


def kitti_segments(gt_poses, est_poses) -> dict:
    """KITTI t_rel (%) and r_rel (deg/m): the per-segment errors of `kitti_segment_table`, each divided by its segment's nominal
    length, averaged over all segments of all lengths (the two figures scripts/benchmarks/_eval_common.py:88-111 reports)."""
    tab = kitti_segment_table(gt_poses, est_poses)
    if len(tab) == 0:
        return {"t_rel_percent": float("nan"), "r_rel_deg_per_m": float("nan")}
    return {"t_rel_percent": float(100.0 * np.mean(tab[:, 3] / tab[:, 2])),
            "r_rel_deg_per_m": float(np.degrees(np.mean(tab[:, 4] / tab[:, 2])))}
