"""Host-side post-processing of the regressor outputs (tiny numpy, as in the reference):
pocolib/utils/poco_utils.py:21-25,50-94 (uncertainty) and pocolib/utils/demo_utils.py:249-281
(camera / keypoint conversion to original-image coordinates)."""
from __future__ import annotations

import numpy as np

from .synth import SMPL_PARENTS


def kinematic_uncert(var: np.ndarray) -> np.ndarray:
    """var[:,child] += var[:,parent] in child order 1..23 (poco_utils.py:21-25)."""
    var = var.copy()
    for i in range(1, 24):
        var[:, i] += var[:, SMPL_PARENTS[i]]
    return var


def prepare_uncert(var, kinematic: bool = True) -> np.ndarray:
    """POCOUtils.prepare_uncert for LOSS_VER norm_flow_* and SIGMA_DIM 1 (poco_utils.py:62-94)."""
    var = np.asarray(var.detach().cpu().numpy() if hasattr(var, "detach") else var, dtype=np.float32)
    if var.ndim == 4:
        var = var.mean(-1).mean(-1)
    elif var.ndim == 3:
        var = var.mean(-1)
    return kinematic_uncert(var) if kinematic else var


def global_uncert(var: np.ndarray, backbone: str, thr: float = 0.40, clip: bool = True) -> np.ndarray:
    """get_global_uncert (poco_utils.py:50-60); folder mode clips it to [0, 0.99] (tester.py:245), video mode
    (tester.py:418-419) does not."""
    var = var.copy()
    if "cliff" in backbone:
        var[var[:, 0] > 2 * thr] = 1.0
        g = var[:, 0]
    else:
        var[var[:, 0] > thr] = 1.0
        g = var.mean(-1)
    return np.clip(g, 0, 0.99) if clip else g


def folder_uncert(var_pose, backbone: str, kinematic: bool):
    """(var, var_global) as POCOTester.run_on_image_folder stores them (tester.py:242-245): kinematic accumulation
    governed by KINEMATIC_UNCERT (= not --no_kinematic_uncert), global value clipped to [0, 0.99]."""
    var = prepare_uncert(var_pose, kinematic)
    return var, global_uncert(var, backbone, clip=True)


def video_uncert(var_pose, backbone: str, kinematic: bool):
    """(var, var_global) as POCOTester.run_on_video stores them (tester.py:416-419).  The `True` the reference passes
    to prepare_uncert there is `return_torch`, NOT a kinematic switch: accumulation is still governed by
    KINEMATIC_UNCERT; the global value is not clipped in this mode."""
    var = prepare_uncert(var_pose, kinematic)
    return var, global_uncert(var, backbone, clip=False)


def convert_crop_cam_to_orig_img(cam, bbox, img_width, img_height):
    cx, cy, h = bbox[:, 0], bbox[:, 1], bbox[:, 2]
    hw, hh = img_width / 2.0, img_height / 2.0
    sx = cam[:, 0] * (1.0 / (img_width / h))
    sy = cam[:, 0] * (1.0 / (img_height / h))
    tx = ((cx - hw) / hw / sx) + cam[:, 1]
    ty = ((cy - hh) / hh / sy) + cam[:, 2]
    return np.stack([sx, sy, tx, ty]).T


def convert_crop_coords_to_orig_img(bbox, keypoints, crop_size):
    cx, cy, h = bbox[:, 0], bbox[:, 1], bbox[:, 2]
    kp = 0.5 * crop_size * (keypoints + 1.0)
    kp = kp * (h[..., None, None] / crop_size)
    kp[:, :, 0] = (cx - h / 2)[..., None] + kp[:, :, 0]
    kp[:, :, 1] = (cy - h / 2)[..., None] + kp[:, :, 1]
    return kp


def write_obj(path, verts, faces=None):
    """Wavefront .obj of one mesh (the reference's --save_obj, tester.py:300-303,532-535: `meshes/<image or person>/
    <idx>.obj`).  verts [V,3] float, faces [F,3] 0-based vertex ids or None (vertex cloud)."""
    import os
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    v = np.asarray(verts, np.float64).reshape(-1, 3)
    with open(path, "w") as f:
        f.write("".join("v %.6f %.6f %.6f\n" % (a, b, c) for a, b, c in v))
        if faces is not None:
            f.write("".join("f %d %d %d\n" % (a + 1, b + 1, c + 1) for a, b, c in np.asarray(faces, np.int64).reshape(-1, 3)))
