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
    """a16: vectors produced by the reference's own get_kinematic_uncert / POCOUtils.prepare_uncert /
    POCOUtils.get_global_uncert (oracle/gen_golden.py run_ops), folder mode (clipped) and video mode (unclipped), with
    and without --no_kinematic_uncert, rows above the 0.8 (cliff) / 0.4 (pare) thresholds included."""
    ops = dict(np.load(GOLD / "ops.npz"))
    assert np.array_equal(postproc.kinematic_uncert(ops["uncert_var"]), ops["uncert_kin"])
    assert np.array_equal(poco_ref.kinematic_uncert(ops["uncert_var"], synth.SMPL_PARENTS), ops["uncert_kin"])
    for bb in ("hrnet_w48_cls-cliff", "hrnet_w32-pare"):
        for kin in (True, False):
            tag = f"{bb.split('-')[1]}_{'kin' if kin else 'nokin'}"
            var, g = postproc.folder_uncert(ops["uncert_var2"], bb, kin)
            assert np.array_equal(var, ops[f"uncert_{tag}_var"]) and np.array_equal(g, ops[f"uncert_{tag}_global"]), tag
            assert g.max() <= 0.99
            var, g = postproc.video_uncert(ops["uncert_var2"], bb, kin)
            assert np.array_equal(var, ops[f"uncert_{tag}_var_video"]), tag
            # video mode runs on torch tensors in the reference: the PARE mean over joints may differ in the last ulp
            assert np.abs(g - ops[f"uncert_{tag}_global_video"]).max() <= 1.2e-7, tag
            assert np.array_equal(poco_ref.global_uncert(var, bb), ops[f"uncert_{tag}_global"])     # the oracle's restatement
    assert ops["uncert_cliff_kin_global_video"].max() == 1.0            # un-clipped in video mode (a thresholded row)
    # video mode must honour --no_kinematic_uncert (the reference's `True` there is return_torch)
    assert not np.array_equal(ops["uncert_cliff_kin_var_video"], ops["uncert_cliff_nokin_var_video"])


def test_packed_record_confidence_is_the_postprocessed_value():
    """dist.pack_records fills the var_global slot with the reference's post-processed scalar."""
    import torch
    from poco_amd import dist as pdist
    ops = dict(np.load(GOLD / "ops.npz"))
    var = torch.from_numpy(ops["uncert_var2"])
    for bb in ("hrnet_w48_cls-cliff", "hrnet_w32-pare"):
        for kin in (True, False):
            tag = f"{bb.split('-')[1]}_{'kin' if kin else 'nokin'}"
            g = pdist.global_confidence(var, bb, kin).numpy()
            assert np.allclose(g, ops[f"uncert_{tag}_global"], atol=1e-7), tag
    out = {"pred_pose": torch.zeros(6, 24, 3, 3), "pred_shape": torch.zeros(6, 10), "pred_cam": torch.zeros(6, 3), "var_pose": var}
    rec = pdist.unpack_records(pdist.pack_records(out, head="hrnet_w48_cls-cliff"))
    assert np.allclose(rec["var_global"].numpy(), ops["uncert_cliff_kin_global"], atol=1e-7)
    assert torch.equal(rec["var_pose"], var)                             # the raw network output travels unchanged


def test_bbox_info_against_reference_vectors():
    """a17: image_utils.calculate_bbox_info / calculate_focal_length outputs recorded from the reference."""
    from poco_amd.tester import calculate_bbox_info, calculate_focal_length
    ops = dict(np.load(GOLD / "ops.npz"))
    for c, s, sh, want, f in zip(ops["bbinfo_center"], ops["bbinfo_scale"], ops["bbinfo_shape"], ops["bbinfo_out"],
                                 ops["bbinfo_focal"]):
        got = calculate_bbox_info(c, s, sh)
        assert got.dtype == np.float32 and np.array_equal(got, want)
        assert calculate_focal_length(sh[0], sh[1]) == f
    # the synthetic batch generator (bench / parity inputs) and the streaming path use the same formula
    info = synth.bbox_info_from(ops["bbinfo_center"], ops["bbinfo_scale"], ops["bbinfo_shape"], ops["bbinfo_focal"])
    assert np.abs(info - ops["bbinfo_out"]).max() < 1e-6
    b = synth.synth_batch(4, 5)
    for i in range(4):
        assert np.allclose(calculate_bbox_info(b["center"][i], b["scale"][i], b["orig_shape"][i]), b["bbox_info"][i], atol=1e-5)


def test_camera_conversions_against_reference_vectors():
    """a18: demo_utils.convert_crop_cam_to_orig_img / convert_crop_coords_to_orig_img outputs recorded from the reference."""
    ops = dict(np.load(GOLD / "ops.npz"))
    oc = postproc.convert_crop_cam_to_orig_img(ops["ccam_cam"], ops["ccam_bbox"], 1920, 1080)
    assert oc.shape == (7, 4) and np.abs(oc - ops["ccam_orig_cam"]).max() <= 1e-6 * np.abs(ops["ccam_orig_cam"]).max()
    kp_in = ops["ccam_kp"].copy()
    kp = postproc.convert_crop_coords_to_orig_img(ops["ccam_bbox"], kp_in, 224)
    assert np.abs(kp - ops["ccam_orig_kp"]).max() <= 1e-6 * np.abs(ops["ccam_orig_kp"]).max()
    assert np.array_equal(kp_in, ops["ccam_kp"])                        # (the product version does not clobber its input)
    # hand-checkable anchor: a centred box maps the crop centre to the image centre
    bbox = np.array([[960.0, 540.0, 300.0, 300.0]], np.float32)
    oc = postproc.convert_crop_cam_to_orig_img(np.array([[0.9, 0.1, -0.2]], np.float32), bbox, 1920, 1080)
    assert np.isclose(oc[0, 0], 0.9 * 300 / 1920) and np.isclose(oc[0, 2], 0.1)


def test_joint_map_and_normalisation_constants_against_reference():
    """constants.JOINT_MAP over constants.JOINT_NAMES (smpl_head.py:17) and IMG_NORM_MEAN/STD (constants.py:2-3)."""
    ops = dict(np.load(GOLD / "ops.npz"))
    assert np.array_equal(ops["joint_map"], synth.JOINT_MAP_49)
    assert np.allclose(ops["img_norm_mean"], [0.485, 0.456, 0.406]) and np.allclose(ops["img_norm_std"], [0.229, 0.224, 0.225])


def test_one_euro_smoothing_against_reference_vectors():
    """oracle/smooth_np.py and the product's vectorised filter vs the reference's OneEuroFilter driven like
    smooth_pose.py:28-61 (tests/golden/smooth.npz, generated by oracle/gen_golden.py)."""
    from oracle import smooth_np
    from poco_amd.smooth import one_euro_rotmats
    g = dict(np.load(GOLD / "smooth.npz"))
    for tag, (mc, be) in {"default": (0.004, 0.7), "demo": (0.004, 1.5)}.items():
        assert np.abs(smooth_np.smooth_rotmats(g["pose"], mc, be) - g[f"hat_{tag}"]).max() < 1e-6
        out = one_euro_rotmats(g["pose"], mc, be)
        assert out.dtype == g["pose"].dtype and np.abs(out - g[f"hat_{tag}"]).max() < 1e-6
    assert np.array_equal(one_euro_rotmats(g["pose"][:1]), g["pose"][:1])          # single frame passes through
    assert one_euro_rotmats(g["pose"][:0]).shape == (0, 24, 3, 3)                  # empty track


def test_reference_cache_files_load(tmp_path):
    """The reference caches its detector / tracker output as joblib pickles (demo.py:125-131,163-169): a per-image
    sequence of [cx,cy,w,h] rows and {person_id: {'bbox','frames'}}; both must load next to the json forms."""
    import json
    import joblib
    from poco_amd.tester import load_detections, load_tracking
    dets = [np.array([[100., 80., 50., 120.]], np.float32), np.zeros((0, 4), np.float32), np.array([[1, 2, 3, 4], [5, 6, 7, 8]], np.float32)]
    joblib.dump(dets, tmp_path / "detection_results.pkl")
    got = load_detections(str(tmp_path / "detection_results.pkl"))
    assert len(got) == 3 and np.array_equal(got[2], dets[2])
    (tmp_path / "d.json").write_text(json.dumps({"a.png": [[1, 2, 3, 4]]}))
    assert load_detections(str(tmp_path / "d.json")) == {"a.png": [[1, 2, 3, 4]]}
    assert load_detections(None) is None
    tr = {1: {"bbox": np.ones((30, 4), np.float32), "frames": np.arange(30)},
          2: {"bbox": np.ones((5, 4), np.float32), "frames": np.arange(5)}}          # < MIN_NUM_FRAMES: dropped
    joblib.dump(tr, tmp_path / "tracking_results_bbox.pkl")
    got = load_tracking(str(tmp_path / "tracking_results_bbox.pkl"))
    assert list(got) == ["1"] and got["1"]["bbox"].shape == (30, 4) and got["1"]["frames"].dtype == np.int64
    (tmp_path / "t.json").write_text(json.dumps({"7": {"bbox": [[1, 2, 3, 4]] * 3, "frames": [0, 1, 2]}}))
    assert load_tracking(str(tmp_path / "t.json"))["7"]["frames"].tolist() == [0, 1, 2]     # json: explicit input, kept


def test_folder_mode_cross_image_batching_host_logic():
    """POCOTester.iter_frame_results (round 3: consecutive images share forwards) with a stub model on the
    CPU: rows go back to the frame they came from, frames without detections yield None in place, a frame with more people
    than the batch is split, forwards are full batches, and at most one batch of crops is pending."""
    import torch
    from poco_amd.tester import POCOTester

    class StubModel:
        max_batch = 4

        def __init__(self):
            self.calls = []
            self.status_checks = 0

        def check_status(self, sync=False):                       # poco_status: asked once per forward, after the D2H copies
            self.status_checks += 1

        def __call__(self, batch, want_segm=False):
            n = batch["img"].shape[0]
            self.calls.append(n)
            tag = batch["img"][:, 0, 0, 0]                        # the crop's tag rides in its first pixel
            return {"tag": tag.clone(), "log_phi": None, "gt_pose_cond_idx": []}

    t = POCOTester.__new__(POCOTester)
    t.model = StubModel()
    t.device = torch.device("cpu")
    t.make_batch = lambda fr, dets, scale: {"img": torch.tensor(np.asarray(dets)[:, 0], dtype=torch.float32).view(-1, 1, 1, 1).repeat(1, 3, 2, 2)}
    t.postprocess = lambda out, dets, W, H: {"tag": out["tag"].numpy(), "cx": np.asarray(dets)[:, 0].astype(np.float32), "wh": np.full(len(dets), W * 1000 + H)}
    people = [1, 0, 3, 9, 0, 2, 1]
    frames, dets, k = [], [], 0
    for i, n in enumerate(people):
        frames.append(np.zeros((10 + i, 20 + i, 3), np.uint8))
        dets.append(np.array([[100 * i + j, 0, 5, 5] for j in range(n)], np.float64).reshape(-1, 4))
    res = t.run_on_frames(frames, dets)
    assert [None if r is None else len(r["tag"]) for r in res] == [1, None, 3, 9, None, 2, 1]
    for i, r in enumerate(res):
        if r is not None:
            assert np.array_equal(r["tag"], np.array([100 * i + j for j in range(people[i])], np.float32))
            assert np.array_equal(r["tag"], r["cx"]) and (r["wh"] == (20 + i) * 1000 + 10 + i).all()
    assert t.model.calls == [4, 4, 4, 4]                       # 16 crops: four full forwards instead of five per-image ones
    assert t.model.status_checks == 4                          # ... each followed by one poco_status query before its rows are used
    # a generator input is consumed lazily: a result comes out as soon as its frame's last crop has been regressed
    t.model.calls.clear()
    seen = []

    def gen():
        for f, d in zip(frames, dets):
            seen.append(len(seen))
            yield f, d

    it = t.iter_frame_results(gen())
    first = next(it)
    assert first is not None and len(seen) <= 4 and t.model.calls == [4]
    assert len([first] + list(it)) == len(frames)
