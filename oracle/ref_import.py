"""ORACLE tooling: import the reference's pocolib.models in THIS container (CPU, no GPU) with
stubbed third-party dependencies.  Only used by oracle/gen_golden.py and by the
optional tests that run when /root/reference exists; nothing here travels to the GPU box.
"""
import logging
import os
import sys
import tempfile
import types

import numpy as np

REFERENCE = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE, "pocolib"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _CfgNode(dict):
    def __init__(self, init=None):
        super().__init__()
        for k, v in (init or {}).items():
            self[k] = _CfgNode(v) if isinstance(v, dict) else v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        import copy
        return copy.deepcopy(self)

    def merge_from_file(self, f):
        raise NotImplementedError

    def freeze(self):
        pass


_workdir = None


def setup(mean_params: dict):
    """Install stubs, chdir into a scratch dir holding data/smpl_mean_params.npz, import pocolib.models."""
    global _workdir
    if "pocolib.models" in sys.modules:
        return sys.modules["pocolib.models"]
    _stub("loguru", logger=logging.getLogger("ref"))
    yacs = _stub("yacs")
    yacs.config = _stub("yacs.config", CfgNode=_CfgNode)
    _stub("flatten_dict", flatten=lambda d, **k: d, unflatten=lambda d, **k: d)
    _stub("pytorch_lightning")
    tv = _stub("torchvision")
    tv.models = _stub("torchvision.models")
    tv.models.utils = _stub("torchvision.models.utils", load_state_dict_from_url=lambda *a, **k: {})

    import torch.nn as nn

    class _SMPL(nn.Module):           # lets `class SMPL(_SMPL)` at smpl_head.py:12 execute
        def __init__(self, *a, **k):
            super().__init__()

    smplx = _stub("smplx", SMPL=_SMPL)
    smplx.body_models = _stub("smplx.body_models", SMPLOutput=dict)
    smplx.lbs = _stub("smplx.lbs", vertices2joints=None)
    _workdir = tempfile.mkdtemp(prefix="poco_ref_")
    os.makedirs(os.path.join(_workdir, "data"))
    np.savez(os.path.join(_workdir, "data", "smpl_mean_params.npz"), **mean_params)
    os.chdir(_workdir)
    sys.path.insert(0, REFERENCE)
    import pocolib.models as M   # noqa
    return M


class _AnyModule(types.ModuleType):
    """A module whose every attribute is another such module / a no-op callable: stands in for third-party packages
    the reference's HOST utilities import at module level but the pinned formulas never call (cv2, trimesh, ...)."""

    def __getattr__(self, k):
        if k.startswith("__"):
            raise AttributeError(k)
        m = _AnyModule(self.__name__ + "." + k)
        sys.modules[m.__name__] = m
        setattr(self, k, m)
        return m

    def __call__(self, *a, **k):
        return None


HOST_STUBS = ["cv2", "trimesh", "trimesh.visual", "trimesh.visual.color", "jpeg4py", "skimage", "skimage.transform",
              "skimage.util", "skimage.util.shape", "pytube", "scipy.misc", "pyrender", "torchvision.transforms",
              "torchvision.utils"]


def setup_host_utils():
    """After setup(): make pocolib.utils.{image_utils,demo_utils,poco_utils} importable so that
    calculate_bbox_info, convert_crop_*_to_orig_img, get_kinematic_uncert and POCOUtils.get_global_uncert /
    prepare_uncert themselves produce the host-formula vectors in tests/golden/ops.npz."""
    import importlib
    import scipy  # noqa: F401  (parent of the scipy.misc stub)
    assert "pocolib.models" in sys.modules, "call setup() first"
    for n in HOST_STUBS:
        if n not in sys.modules:
            sys.modules[n] = _AnyModule(n)
            if "." in n:
                par = sys.modules.get(n.rsplit(".", 1)[0])
                if par is not None:
                    setattr(par, n.rsplit(".", 1)[1], sys.modules[n])
    mods = {}
    for name in ("image_utils", "demo_utils", "poco_utils"):
        mods[name] = importlib.import_module(f"pocolib.utils.{name}")
    mods["constants"] = importlib.import_module("pocolib.core.constants")
    return mods


def poco_utils_instance(mods, backbone: str, kinematic: bool):
    """POCOUtils(hparams) as POCOTester builds it (tester.py:59-60) for the demo configs."""
    hp = _CfgNode({"PL_LOGGING": False, "PREF_LOGGER": "none", "METHOD": "demo",
                   "POCO": {"LOSS_VER": "norm_flow_res_gaus", "BACKBONE": backbone, "UNCERT_TYPE": ["pose"],
                            "KINEMATIC_UNCERT": kinematic, "LOG_UNCERT_STAT": False, "EXCLUDE_UNCERT_IDX": ""}})
    return mods["poco_utils"].POCOUtils(hp)
