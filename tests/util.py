"""Shared helpers for the parity tests (seeded synthetic weights -> engine / oracle)."""
import json
from pathlib import Path

import numpy as np

from poco_amd import synth

GOLD = Path(__file__).parent / "golden"
FLOW_LAYERS = {"hrnet_w32-pare": 3, "hrnet_w48_cls-cliff": 1, "resnet50-cliff": 1}


def load_spec(variant):
    return [(n, tuple(s)) for n, s in json.loads((GOLD / f"spec_{variant}.json").read_text())]


def synth_weights(variant, seed=0, profile="default"):
    """profile "stress": every BN gamma in [0.5,1.5] with calibrated running statistics (poco_amd/synth.py)."""
    calib = synth.load_calib(variant) if profile == "stress" else None
    w = synth.synth_state_dict(load_spec(variant), seed, profile, calib)
    return {k: v for k, v in w.items() if v.dtype != np.int64}


def make_engine(variant, max_batch, seed=0, smpl_seed=7, profile="default", options=None):
    """options: poco_create_ex build options (dict), e.g. {"kmerge": 0} for the separate-launch form of an op group."""
    from poco_amd.model import POCO
    m = POCO(backbone=variant, num_flow_layers=FLOW_LAYERS[variant], max_batch=max_batch, smpl=synth.synth_smpl(smpl_seed),
             keep_state_dict=False, engine_options=options)
    m.load_state_dict(synth_weights(variant, seed, profile), strict=True)
    return m.finalize()


def cuda_batch(batch_np, device):
    import torch
    return {k: torch.from_numpy(v).to(device) for k, v in batch_np.items()}


def oracle_forward(variant, batch_np, seed=0, smpl_seed=7, profile="default"):
    from oracle import poco_ref
    sd = poco_ref.to_torch(synth_weights(variant, seed, profile))
    return poco_ref.poco_forward(variant, sd, poco_ref.to_torch(synth.synth_smpl(smpl_seed)), poco_ref.to_torch(batch_np))


def has_experiments() -> bool:
    """Is the loaded library an experiment build (python -m poco_amd.build --experiments; POCO_HIP_LIB=poco_amd/lib/exp/libpoco_hip_experiments.so)?
    The shipped library contains neither the split-fp16 GEMM (ALG 12) nor the 3-deep rings of ALG 4: `split_f16` is then an unknown
    build option (host-only check, no GPU needed)."""
    import ctypes as C
    from poco_amd import _lib
    L = _lib.lib()
    L.poco_create_ex.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.POINTER(C.c_void_p)]
    L.poco_destroy.argtypes = [C.c_void_p]
    L.poco_destroy.restype = None
    h = C.c_void_p()
    if L.poco_create_ex(b"resnet50-cliff", 1, 1, b"split_f16=1", C.byref(h)) != 0:
        return False
    L.poco_destroy(h)
    return True
