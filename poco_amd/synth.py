"""Seeded synthetic weights / SMPL model / input batches.

The reference's checkpoints (data/poco_*.pt) and the SMPL body model are license-gated and not in
the tree, and there is no network.  Everything that needs numbers - bench.py,
the parity tests, the golden-vector generator - therefore draws them from here: numpy PCG64
streams keyed by (seed, crc32(tensor name)), so the same tensors come out on every machine and
independent of iteration order.

Scale rules keep ~300 stacked conv/BN/ReLU layers O(1) (the reference's own init, std=0.001 at
pocolib/models/backbone/hrnet.py:535, collapses activations to 0 and would make parity vacuous).

Two weight profiles:
  "default"  every BN gamma in [0.8,1.2] but the last BN of a residual branch / fuse path damped
             (gamma ~0.1) so that un-normalised stacks stay O(1).  Used by bench.py and the tuner
             (timing does not depend on the values).
  "stress"   SURVEY.md 8(c): EVERY BN gamma in [0.5,1.5], beta in [-0.2,0.2]; the residual branches
             carry as much signal as the trunk.  Activations stay O(1..10) the way they do in a
             trained network: each BN's running statistics track its input.  The per-layer scalars
             (mean, variance of the conv output feeding each BN) were measured once, layer by layer,
             on a seeded calibration batch through the reference's own modules
             (oracle/gen_golden.py calibrate) and are committed as data in poco_amd/calib/.
             running_mean_c = m + sqrt(v)*U(-0.2,0.2), running_var_c = v*U(0.5,1.5).
             Decoder / MLP gains are raised too, so pose/shape/confidence move O(0.1..1) between crops.
"""
from __future__ import annotations

import zlib
import json
from pathlib import Path
from typing import Dict, Iterable, Optional, Sequence, Tuple

import numpy as np

CALIB_DIR = Path(__file__).resolve().parent / "calib"

Spec = Sequence[Tuple[str, Tuple[int, ...]]]


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def alter_masks(num_rv: int, num_flow_layers: int) -> np.ndarray:
    """Alternating RealNVP masks, semantics of pocolib/models/head/nf_head.py:20-21."""
    a = [i % 2 for i in range(num_rv)]
    b = [(i + 1) % 2 for i in reversed(range(num_rv))]
    return np.array([a, b] * num_flow_layers, dtype=np.float32)


def load_calib(variant: str) -> Dict[str, Tuple[float, float]]:
    """BN name -> (mean, variance) of its input on the calibration batch ("stress" profile)."""
    f = CALIB_DIR / f"stress_{variant}.json"
    if not f.exists():
        raise FileNotFoundError(f"{f}: run oracle/gen_golden.py calibrate (needs the reference)")
    return {k: (float(m), float(v)) for k, (m, v) in json.loads(f.read_text()).items()}


STRESS_LINEAR_GAIN = {"decpose": 0.25, "decshape": 0.25, "deccam": 0.08, "fc1": 1.0, "fc2": 1.0,
                      "cam_mlp": 0.1, "shape_mlp": 1.0, "uncert_fc_featNet": 2.0, "uncert_fc_poseNet": 2.0,
                      "uncert_fc1": 3.0, "uncert_fc2": 3.0}
DEFAULT_LINEAR_GAIN = {"decpose": 0.05, "decshape": 0.05, "deccam": 0.05, "fc1": 0.5, "fc2": 0.5,
                       "cam_mlp": 0.1, "shape_mlp": 0.5}


def synth_state_dict(spec: Spec, seed: int = 0, profile: str = "default",
                     calib: Optional[Dict[str, Tuple[float, float]]] = None) -> Dict[str, np.ndarray]:
    """profile "stress": `calib` maps BN module names to the (mean, var) of their input; a BN without
    an entry gets (0, 1) - that is how the calibration pass itself starts."""
    if profile not in ("default", "stress"):
        raise ValueError(profile)
    stress = profile == "stress"
    calib = calib or {}
    gains = STRESS_LINEAR_GAIN if stress else DEFAULT_LINEAR_GAIN
    names = {n for n, _ in spec}
    out: Dict[str, np.ndarray] = {}
    for name, shape in spec:
        shape = tuple(int(s) for s in shape)
        r = _rng(seed, name)
        stem, _, leaf = name.rpartition(".")
        is_bn = (stem + ".running_mean") in names
        if leaf == "num_batches_tracked":
            out[name] = np.zeros(shape, dtype=np.int64)
        elif is_bn:
            m0, v0 = calib.get(stem, (0.0, 1.0)) if stress else (0.0, 1.0)
            if leaf == "weight" and stress:
                v = r.uniform(0.5, 1.5, shape)
            elif leaf == "weight":
                lo, hi = 0.8, 1.2
                last = stem.rsplit(".", 1)[-1]
                parent = stem.rsplit(".", 1)[0]
                if "fuse_layers" in stem:
                    lo, hi = 0.10, 0.20
                elif last == "bn3" or (last == "bn2" and (parent + ".bn3.weight") not in names
                                       and (parent + ".conv2.weight") in names):
                    lo, hi = 0.05, 0.15       # last BN of a residual branch
                v = r.uniform(lo, hi, shape)
            elif leaf == "bias":
                v = r.uniform(-0.2, 0.2, shape)
            elif leaf == "running_mean":
                v = m0 + np.sqrt(v0) * r.uniform(-0.2, 0.2, shape)
            elif leaf == "running_var":
                v = v0 * r.uniform(0.5, 1.5, shape)
            else:
                raise ValueError(name)
            out[name] = v.astype(np.float32)
        elif name.endswith("flow.mask"):
            out[name] = alter_masks(shape[1], shape[0] // 2)
        elif leaf == "temperature":
            out[name] = np.ones(shape, dtype=np.float32)
        elif leaf == "init_pose":
            ident = np.tile(np.array([1, 0, 0, 1, 0, 0], dtype=np.float64), shape[-1] // 6)
            out[name] = (ident + 0.1 * r.standard_normal(shape[-1])).reshape(shape).astype(np.float32)
        elif leaf == "init_shape":
            out[name] = (0.1 * r.standard_normal(shape)).astype(np.float32)
        elif leaf == "init_cam":
            out[name] = np.array([0.9, 0.0, 0.0], dtype=np.float32).reshape(shape)
        elif leaf == "weight" and len(shape) == 4:          # conv OIHW
            fan_in = shape[1] * shape[2] * shape[3]
            out[name] = (r.standard_normal(shape) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        elif leaf == "weight" and len(shape) == 6:          # LocallyConnected2d [1,O,C,J,1,1]
            out[name] = (r.standard_normal(shape) / np.sqrt(shape[2])).astype(np.float32)
        elif leaf == "weight" and len(shape) == 2:          # Linear [out,in]
            gain = gains.get(stem.rsplit(".", 1)[-1], 1.0)
            out[name] = (gain * r.standard_normal(shape) / np.sqrt(shape[1])).astype(np.float32)
        elif leaf == "bias":
            v = r.uniform(-0.05, 0.05, shape)
            if stem.rsplit(".", 1)[-1] == "cam_mlp":
                v = v + np.array([0.9, 0.0, 0.0])
            out[name] = v.astype(np.float32)
        else:
            raise ValueError(f"no synthetic rule for tensor {name} {shape}")
    return out


# --------------------------------------------------------------------------------------------
# SMPL-shaped body model (NOT the real SMPL: same tensor shapes/sparsity pattern, random content)
# --------------------------------------------------------------------------------------------
SMPL_PARENTS = np.array([-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21],
                        dtype=np.int32)   # pocolib/utils/kp_utils.py:881-908
NUM_VERTS = 6890
# 21 extra joints picked from vertices by smplx's VertexJointSelector ('smplh' ids; recalled from
# the public smplx table, see SURVEY.md 3.5 - kept as data so it can be corrected without a rebuild)
SMPL_EXTRA_VERTEX_IDS = np.array(
    [332, 6260, 2800, 4071, 583, 3216, 3226, 3387, 6617, 6624, 6787,
     2746, 2319, 2445, 2556, 2673, 6191, 5782, 5905, 6016, 6133], dtype=np.int32)
# constants.JOINT_MAP order over the 54 = 24 + 21 + 9 joints (pocolib/core/constants.py:15-91)
JOINT_MAP_49 = np.array(
    [24, 12, 17, 19, 21, 16, 18, 20, 0, 2, 5, 8, 1, 4, 7, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34,
     8, 5, 45, 46, 4, 7, 21, 19, 17, 16, 18, 20, 47, 48, 49, 50, 51, 52, 53, 24, 26, 25, 28, 27],
    dtype=np.int32)


def synth_smpl(seed: int = 7, num_verts: int = NUM_VERTS) -> Dict[str, np.ndarray]:
    r = np.random.default_rng(seed)
    V = num_verts
    v_template = np.stack([r.uniform(-0.45, 0.45, V), r.uniform(-1.1, 0.7, V), r.uniform(-0.15, 0.15, V)], 1)
    shapedirs = 0.02 * r.standard_normal((V, 3, 10))
    posedirs = 0.002 * r.standard_normal((207, V * 3))

    def sparse_rows(rows, nnz):
        m = np.zeros((rows, V))
        for i in range(rows):
            idx = r.choice(V, nnz, replace=False)
            m[i, idx] = r.dirichlet(np.ones(nnz))
        return m

    J_regressor = sparse_rows(24, 32)
    J_regressor_extra = sparse_rows(9, 16)
    lbs_weights = np.zeros((V, 24))
    for i in range(V):
        idx = r.choice(24, 4, replace=False)
        lbs_weights[i, idx] = r.dirichlet(np.ones(4))
    return {
        "v_template": v_template.astype(np.float32),
        "shapedirs": shapedirs.astype(np.float32),
        "posedirs": posedirs.astype(np.float32),
        "J_regressor": J_regressor.astype(np.float32),
        "J_regressor_extra": J_regressor_extra.astype(np.float32),
        "lbs_weights": lbs_weights.astype(np.float32),
        "parents": SMPL_PARENTS.copy(),
        "extra_vertex_ids": np.minimum(SMPL_EXTRA_VERTEX_IDS, V - 1).astype(np.int32),
        "joint_map": JOINT_MAP_49.copy(),
    }


# --------------------------------------------------------------------------------------------
# Input batches (contract: pocolib/core/tester.py:178-212, pocolib/utils/image_utils.py:171-187)
# --------------------------------------------------------------------------------------------
def bbox_info_from(center: np.ndarray, scale: np.ndarray, orig_shape: np.ndarray, focal: np.ndarray) -> np.ndarray:
    """[(cx-W/2)/f*2.8, (cy-H/2)/f*2.8, (b-0.24f)/(0.06f)], b = scale*200."""
    img_h, img_w = orig_shape[:, 0], orig_shape[:, 1]
    b = scale * 200.0
    info = np.stack([center[:, 0] - img_w / 2.0, center[:, 1] - img_h / 2.0, b], -1)
    info[:, :2] = info[:, :2] / focal[:, None] * 2.8
    info[:, 2] = (info[:, 2] - 0.24 * focal) / (0.06 * focal)
    return info.astype(np.float32)


def synth_batch(B: int, seed: int = 1234, img_w: int = 1920, img_h: int = 1080,
                profile: str = "default") -> Dict[str, np.ndarray]:
    """profile "default": white-noise crops.  profile "stress": every crop additionally carries
    its own low-frequency pattern (a random 7x7x3 grid, x32 nearest), contrast and colour offset, so that globally
    pooled features - and with them cam / confidence - differ between crops by far more than the parity gate."""
    r = np.random.default_rng(seed)
    img = r.standard_normal((B, 3, 224, 224), dtype=np.float32)
    if profile == "stress":
        rs = np.random.default_rng([seed, 1])
        grid = rs.standard_normal((B, 3, 7, 7)).astype(np.float32)
        contrast = rs.uniform(0.4, 2.0, (B, 1, 1, 1)).astype(np.float32)
        offset = rs.uniform(-0.7, 0.7, (B, 3, 1, 1)).astype(np.float32)
        img = contrast * (0.6 * img + 0.8 * grid.repeat(32, axis=2).repeat(32, axis=3)) + offset
    elif profile != "default":
        raise ValueError(profile)
    center = np.array([img_w / 2.0, img_h / 2.0]) + r.uniform(-0.25, 0.25, (B, 2)) * np.array([img_w, img_h])
    side = r.uniform(150.0, 600.0, B)
    scale = side / 200.0
    focal = np.full(B, np.sqrt(img_w ** 2 + img_h ** 2))
    orig_shape = np.tile(np.array([[img_h, img_w]], dtype=np.float64), (B, 1))
    return {
        "img": img,
        "bbox_info": bbox_info_from(center, scale, orig_shape, focal),
        "focal_length": focal.astype(np.float32),
        "scale": scale.astype(np.float32),
        "center": center.astype(np.float32),
        "orig_shape": orig_shape.astype(np.float32),
    }
