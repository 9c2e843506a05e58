"""Pose-chain helpers between the front-end's output and its evaluation: what the reference
keeps in utils/utils.py (lines 17-141: relative_chain_to_absolute, centers_from_Rts,
chain_rel_to_abs_w2c, rot_err_deg, rel_position_err, compute_position_errs, to_hom, invert_T,
relatives_from_abs_w2c_endpoints, relatives_from_absolutes_w2c_adjacent) under the same names
and conventions, plus the RMSE its trajectory plot prints (utils/utils.py:165-169).  Host-side
NumPy, as there: a few hundred 4x4 products per sequence.

All poses are world -> camera [R|t]; `relative_Rts[k-1]` (what the front-end writes, and the
reference pickles, fill_database.py:252-255) takes frame k-1 coordinates to frame k.
"""
from __future__ import annotations

import numpy as np


def to_hom(T34: np.ndarray) -> np.ndarray:
    """(N,3,4) -> (N,4,4)."""
    T34 = np.asarray(T34)
    out = np.zeros((T34.shape[0], 4, 4), dtype=T34.dtype)
    out[:, :3, :] = T34
    out[:, 3, 3] = 1
    return out


def invert_T(T: np.ndarray) -> np.ndarray:
    """Inverses of rigid (N,4,4) transforms: [R^T | -R^T t]."""
    T = np.asarray(T)
    Rt = np.swapaxes(T[:, :3, :3], 1, 2)
    out = np.zeros_like(T)
    out[:, :3, :3] = Rt
    out[:, :3, 3] = -(Rt @ T[:, :3, 3:4])[:, :, 0]
    out[:, 3, 3] = 1
    return out


def _compose_chain(steps: np.ndarray) -> np.ndarray:
    """T_cum[i] = steps[i] @ ... @ steps[0] for (N,3,4) steps, as (N,3,4)."""
    out = np.empty_like(steps)
    R, t = np.eye(3), np.zeros(3)
    for i in range(steps.shape[0]):
        Rs, ts = steps[i, :, :3], steps[i, :, 3]
        R, t = Rs @ R, Rs @ t + ts
        out[i, :, :3], out[i, :, 3] = R, t
    return out


def relative_chain_to_absolute(Rts_rel: np.ndarray) -> np.ndarray:
    """(N,3,4) consecutive T_{i,i-1} -> (N,3,4) cumulative T_{i,0} (the first entry is the
    first step itself: world = the camera before it)."""
    return _compose_chain(np.asarray(Rts_rel))


def chain_rel_to_abs_w2c(rel_Rts: np.ndarray) -> np.ndarray:
    """(N-1,3,4) relative poses -> (N,3,4) world -> camera poses with world = frame 0."""
    rel_Rts = np.asarray(rel_Rts, dtype=np.float64)
    out = np.zeros((rel_Rts.shape[0] + 1, 3, 4))
    out[0, :, :3] = np.eye(3)
    if rel_Rts.shape[0]:
        out[1:] = _compose_chain(rel_Rts)
    return out


def centers_from_Rts(Rts_w2c: np.ndarray) -> np.ndarray:
    """Camera centres C = -R^T t of (N,3,4) world -> camera poses."""
    Rts_w2c = np.asarray(Rts_w2c)
    return -(np.swapaxes(Rts_w2c[:, :3, :3], 1, 2) @ Rts_w2c[:, :3, 3:4])[:, :, 0]


def rot_err_deg(T_est: np.ndarray, T_gt: np.ndarray) -> np.ndarray:
    """Per-frame angle (degrees) of R_gt^T R_est: the norm of its rotation vector, which is
    what the reference takes from cv2.Rodrigues."""
    R = np.swapaxes(np.asarray(T_gt)[:, :3, :3], 1, 2) @ np.asarray(T_est)[:, :3, :3]
    # angle from both the trace and the skew part: accurate near 0 and near 180 degrees
    skew = np.stack([R[:, 2, 1] - R[:, 1, 2], R[:, 0, 2] - R[:, 2, 0], R[:, 1, 0] - R[:, 0, 1]], axis=1)
    s = 0.5 * np.linalg.norm(skew, axis=1)
    c = 0.5 * (np.trace(R, axis1=1, axis2=2) - 1.0)
    return np.degrees(np.arctan2(s, c))


def rel_position_err(T_rel: np.ndarray, T_gt_rel: np.ndarray) -> np.ndarray:
    return np.linalg.norm(np.asarray(T_gt_rel)[:, :3, 3] - np.asarray(T_rel)[:, :3, 3], axis=1)


def compute_position_errs(T_est_w2c: np.ndarray, T_gt_w2c: np.ndarray) -> dict:
    """Absolute camera-centre error per frame: |dx|, |dy|, |dz| and its norm."""
    d = centers_from_Rts(T_est_w2c) - centers_from_Rts(T_gt_w2c)
    a = np.abs(d)
    return dict(ex=a[:, 0], ey=a[:, 1], ez=a[:, 2], epos=np.linalg.norm(d, axis=1))


def relatives_from_abs_w2c_endpoints(T_abs: np.ndarray, starts, ends) -> np.ndarray:
    """T_{starts <- ends} = T_abs[starts] @ inv(T_abs[ends]), homogeneous (M,4,4)."""
    Th = to_hom(np.asarray(T_abs))
    return Th[starts] @ invert_T(Th[ends])


def relatives_from_absolutes_w2c_adjacent(abs_poses: np.ndarray, KF_indices=None) -> np.ndarray:
    """T_{k <- k+1} between consecutive poses (or consecutive keyframes), homogeneous."""
    Th = to_hom(np.asarray(abs_poses))
    if KF_indices is not None:
        Th = Th[np.asarray(KF_indices)]
    return Th[:-1] @ invert_T(Th[1:])


def trajectory_rmse(T_est_w2c: np.ndarray, T_gt_w2c: np.ndarray) -> float:
    """sqrt(mean |C_est - C_gt|^2): the figure the reference prints beside each trajectory
    (utils/utils.py:165-169; README: 27.00 / 8.60 / 2.39 m after PnP / BA / LC on KITTI-05)."""
    d = centers_from_Rts(T_est_w2c) - centers_from_Rts(T_gt_w2c)
    return float(np.sqrt(np.mean(np.sum(d * d, axis=1))))
