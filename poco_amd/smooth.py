"""Temporal smoothing of a tracklet (pocolib/utils/smooth_pose.py:25-69, demo flag --smooth).

The reference filters every entry of the [24,3,3] rotation matrices with a One Euro filter
(pocolib/utils/one_euro_filter.py) frame by frame and calls the SMPL layer once PER FRAME on the CPU.
Here the filter recurrence (inherently sequential in time, 216 independent channels) runs vectorised on the
host in one pass, and the T meshes come from ONE batched call of the HIP LBS operator (poco_smpl_lbs).
"""
from __future__ import annotations

import math
from typing import Tuple

import numpy as np


def one_euro_rotmats(pose: np.ndarray, min_cutoff: float = 0.004, beta: float = 0.7, d_cutoff: float = 1.0) -> np.ndarray:
    """pose [T,24,3,3] -> smoothed [T,24,3,3].  Unit frame spacing (the reference feeds t = frame index),
    frame 0 passes through (smooth_pose.py:40-41)."""
    pose = np.asarray(pose)
    T = pose.shape[0]
    out = np.empty_like(pose)
    if T == 0:
        return out
    out[0] = pose[0]
    r_d = 2.0 * math.pi * d_cutoff            # t_e = 1
    a_d = r_d / (r_d + 1.0)
    x_prev = pose[0].astype(pose.dtype)
    dx_prev = np.zeros_like(x_prev)
    for i in range(1, T):
        x = pose[i]
        dx_hat = a_d * (x - x_prev) + (1.0 - a_d) * dx_prev
        r = 2.0 * math.pi * (min_cutoff + beta * np.abs(dx_hat))
        a = r / (r + 1.0)
        x_prev = a * x + (1.0 - a) * x_prev
        dx_prev = dx_hat
        out[i] = x_prev
    return out


def smooth_pose(model, pred_pose: np.ndarray, pred_betas: np.ndarray, min_cutoff: float = 0.004,
                beta: float = 0.7) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """-> (verts [T,6890,3], pose_hat [T,24,3,3], joints3d [T,49,3]); `model` is a finalized poco_amd.model.POCO.
    The smoothed matrices are NOT re-orthogonalised (neither does the reference)."""
    import torch
    pose_hat = one_euro_rotmats(pred_pose, min_cutoff, beta)
    T = pose_hat.shape[0]
    verts = np.empty((T, 6890, 3), np.float32)
    joints = np.empty((T, 49, 3), np.float32)
    bs = model.max_batch
    for lo in range(0, T, bs):
        hi = min(T, lo + bs)
        b = torch.from_numpy(np.ascontiguousarray(pred_betas[lo:hi], np.float32)).to(model.device)
        R = torch.from_numpy(np.ascontiguousarray(pose_hat[lo:hi], np.float32)).to(model.device)
        v, j = model.smpl_lbs(b, R)
        verts[lo:hi] = v.cpu().numpy()
        joints[lo:hi] = j.cpu().numpy()
    return verts, pose_hat, joints
