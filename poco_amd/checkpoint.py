"""Checkpoint reading (the READ side of pocolib/models/poco.py:131-154 and
pocolib/utils/train_utils.py:69-136): accepts the reference's .pt/.ckpt/.pth files or run
directories and returns {part-prefixed key -> numpy fp32} for POCO.load_state_dict.

Differences from the reference, on purpose: loading is strict (the reference silently falls back
to strict=False, train_utils.py:118-124) and only the four model parts are kept."""
from __future__ import annotations

import glob
from typing import Dict

import numpy as np

PARTS = ("backbone", "head", "uncert_head", "flow_head")


def get_model_path(path: str, inf_model: str = "best") -> str:
    """train_utils.py:126-136."""
    if path.endswith((".pt", ".ckpt", ".pth")):
        return path
    if inf_model == "best":
        return path + "/best_model.pt"
    if inf_model == "best_mpjpe_var":
        return path + "/best_mpjpe_var_model.pt"
    return sorted(glob.glob(f"{path}/tb_logs_poco-smpl/*/checkpoints/*"))[-1]


def split_parts(state_dict: Dict[str, object]) -> Dict[str, np.ndarray]:
    """Keep keys of the four parts, strip a leading 'model.' (train_utils.py:69-90)."""
    out = {}
    for k, v in state_dict.items():
        k2 = k[len("model."):] if k.startswith("model.") else k
        if not any(k2.startswith(p + ".") for p in PARTS):
            continue
        arr = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        if arr.dtype.kind in "iu":           # num_batches_tracked: tolerated, not needed
            continue
        out[k2] = np.ascontiguousarray(arr, dtype=np.float32)
    return out


def read_checkpoint(file: str, inf_model: str = "best") -> Dict[str, np.ndarray]:
    import torch
    path = get_model_path(str(file), inf_model)
    sd = torch.load(path, map_location="cpu", weights_only=False)
    sd = sd["model"] if "model" in sd else sd
    sd = sd["state_dict"] if "state_dict" in sd else sd
    return split_parts(sd)
