"""CPU: host-side logic that mirrors the reference's plumbing (config schema, CLI flags, uncertainty
post-processing, camera conversions, bbox_info)."""
import numpy as np
import pytest

from oracle import poco_ref
from poco_amd import postproc, synth
from poco_amd.config import model_kwargs, update_hparams
from tests.util import GOLD


def test_yaml_schema_and_model_kwargs():
    hp = update_hparams("configs/demo_poco_pare.yaml")
    kw = model_kwargs(hp)
    assert kw["backbone"] == "hrnet_w32-pare" and kw["num_flow_layers"] == 3 and kw["num_neurons"] == "512-"
    hp = update_hparams("configs/demo_poco_cliff.yaml")
    assert model_kwargs(hp)["uncert_inp_type"] == "feat-pose-net" and hp.DATASET.IMG_RES == 224


def test_cli_flags_match_reference_semantics():
    import demo
    a = demo.parse_args(["--cfg", "c.yaml", "--ckpt", "m.pt"])
    assert a.batch_size == 64 and a.mode == "folder" and a.no_kinematic_uncert is True   # store_false default
    a = demo.parse_args(["--cfg", "c.yaml", "--ckpt", "m.pt", "--no_kinematic_uncert", "--mode", "video"])
    assert a.no_kinematic_uncert is False and a.mode == "video"


def test_uncert_postproc_against_reference_vectors():
    ops = dict(np.load(GOLD / "ops.npz"))
    assert np.array_equal(postproc.kinematic_uncert(ops["uncert_var"]), ops["uncert_kin"])
    var = postproc.prepare_uncert(ops["uncert_var"], kinematic=True)
    for bb in ("hrnet_w48_cls-cliff", "hrnet_w32-pare"):
        g = postproc.global_uncert(var, bb)
        assert np.allclose(g, poco_ref.global_uncert(var, bb)) and g.max() <= 0.99


def test_bbox_info_matches_batch_generator():
    from poco_amd.tester import calculate_bbox_info, calculate_focal_length
    b = synth.synth_batch(4, 5)
    for i in range(4):
        info = calculate_bbox_info(b["center"][i], b["scale"][i], b["orig_shape"][i])
        assert np.allclose(info, b["bbox_info"][i], atol=1e-5)
    assert abs(calculate_focal_length(1080, 1920) - 2202.9071) < 1e-3


def test_camera_conversions():
    cam = np.array([[0.9, 0.1, -0.2]], np.float32)
    bbox = np.array([[960.0, 540.0, 300.0, 300.0]], np.float32)
    oc = postproc.convert_crop_cam_to_orig_img(cam, bbox, 1920, 1080)
    assert oc.shape == (1, 4) and np.isclose(oc[0, 0], 0.9 * 300 / 1920) and np.isclose(oc[0, 2], 0.1)
    kp = postproc.convert_crop_coords_to_orig_img(bbox, np.zeros((1, 49, 2), np.float32), 224)
    assert np.allclose(kp[0, :, 0], 960.0) and np.allclose(kp[0, :, 1], 540.0)
