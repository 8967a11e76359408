"""ORACLE (test infrastructure only): temporal smoothing of a tracklet's SMPL pose, restating
pocolib/utils/one_euro_filter.py:20-61 (One Euro filter on every rotation-matrix entry) as driven by
pocolib/utils/smooth_pose.py:25-69: t0 = 0, x0 = pose[0], frame idx uses t = idx (so t_e = 1 for every
step), frame 0 is passed through; vertices / joints are the SMPL LBS of the smoothed pose with the
frame's own betas.  Pinned against the reference's OneEuroFilter class by oracle/gen_golden.py
(tests/golden/smooth.npz).
"""
import math

import numpy as np


def smoothing_factor(t_e, cutoff):            # one_euro_filter.py:20-22
    r = 2 * math.pi * cutoff * t_e
    return r / (r + 1)


def smooth_rotmats(pose: np.ndarray, min_cutoff=0.004, beta=0.7, d_cutoff=1.0) -> np.ndarray:
    """pose [T,24,3,3] -> filtered [T,24,3,3] (smooth_pose.py:28-35,57-61; one_euro_filter.py:43-61)."""
    out = np.zeros_like(pose)
    out[0] = pose[0]
    x_prev = pose[0]
    dx_prev = 0.0
    t_prev = np.zeros_like(pose[0])
    for idx in range(1, pose.shape[0]):
        t = np.ones_like(pose[idx]) * idx
        t_e = t - t_prev
        a_d = smoothing_factor(t_e, d_cutoff)
        dx = (pose[idx] - x_prev) / t_e
        dx_hat = a_d * dx + (1 - a_d) * dx_prev
        cutoff = min_cutoff + beta * np.abs(dx_hat)
        a = smoothing_factor(t_e, cutoff)
        x_hat = a * pose[idx] + (1 - a) * x_prev
        x_prev, dx_prev, t_prev = x_hat, dx_hat, t
        out[idx] = x_hat
    return out


def smooth_pose(pose, betas, smpl, min_cutoff=0.004, beta=0.7):
    """-> (verts [T,6890,3], pose_hat [T,24,3,3], joints49 [T,49,3]) like smooth_pose.py:25-69."""
    from . import smpl_np
    pose_hat = smooth_rotmats(pose, min_cutoff, beta)
    verts, joints = smpl_np.smpl_lbs_np(smpl, betas, pose_hat)
    return verts, pose_hat, joints
