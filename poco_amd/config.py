"""The slice of pocolib/core/config.py the inference path reads: the yaml schema of
configs/demo_poco_{pare,cliff}.yaml (METHOD + the POCO: map) with the reference's defaults
(config.py:84-229) for keys a yaml may omit.  Plain PyYAML; no yacs."""
from __future__ import annotations

from types import SimpleNamespace

import yaml

POCO_DEFAULTS = dict(
    BACKBONE="resnet50-cliff", UNCERT_LAYER="diff_branch", ACTIVATION_TYPE="sigmoid", UNCERT_TYPE="pose",
    UNCERT_INP_TYPE="feat", LOSS_VER="norm_flow_res_gaus", NUM_NEURONS="1024-512", NUM_FLOW_LAYERS=3, SIGMA_DIM=1,
    NUM_NF_RV=9, MASK_PARAMS_ID="", NFLOW_MASK_TYPE="alter", EXCLUDE_UNCERT_IDX="", USE_DROPOUT=False,
    USE_ITER_FEATS=False, COND_NFLOW=True, CONTEXT_DIM=512, GT_POSE_COND=False, GT_POSE_COND_DS="h36m",
    GT_POSE_COND_RATIO=0.25, KINEMATIC_UNCERT=False)


def update_hparams(cfg_file: str) -> SimpleNamespace:
    raw = yaml.safe_load(open(cfg_file)) or {}
    poco = dict(POCO_DEFAULTS)
    poco.update(raw.get("POCO", {}) or {})
    return SimpleNamespace(METHOD=raw.get("METHOD", "poco"), POCO=SimpleNamespace(**poco),
                           DATASET=SimpleNamespace(IMG_RES=(raw.get("DATASET", {}) or {}).get("IMG_RES", 224)))


def model_kwargs(hp: SimpleNamespace) -> dict:
    """Keyword arguments of POCO(...) exactly as pocolib/core/tester.py:75-98 passes them."""
    p = hp.POCO
    return dict(backbone=p.BACKBONE, img_res=hp.DATASET.IMG_RES, uncert_layer=p.UNCERT_LAYER,
                activation_type=p.ACTIVATION_TYPE, uncert_type=p.UNCERT_TYPE, uncert_inp_type=p.UNCERT_INP_TYPE,
                loss_ver=p.LOSS_VER, num_neurons=p.NUM_NEURONS, num_flow_layers=p.NUM_FLOW_LAYERS, sigma_dim=p.SIGMA_DIM,
                num_nf_rv=p.NUM_NF_RV, mask_params_id=p.MASK_PARAMS_ID, nflow_mask_type=p.NFLOW_MASK_TYPE,
                exclude_uncert_idx=p.EXCLUDE_UNCERT_IDX, use_dropout=p.USE_DROPOUT, use_iter_feats=p.USE_ITER_FEATS,
                cond_nflow=p.COND_NFLOW, context_dim=p.CONTEXT_DIM, gt_pose_cond=p.GT_POSE_COND,
                gt_pose_cond_ratio=p.GT_POSE_COND_RATIO)
