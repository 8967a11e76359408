"""GPU: the crop kernel against its numpy oracle, and demo.py end to end (folder mode) on a synthetic
checkpoint / SMPL file / images, cross-checked against the CPU oracle on the very same crops."""
import json

import numpy as np
import pytest
import torch

from poco_amd import synth
from tests import util

pytestmark = pytest.mark.gpu


def test_crop_normalize_kernel(cuda):
    from oracle.crop_np import crop_normalize_np
    from poco_amd.tester import crop_normalize
    r = np.random.default_rng(3)
    frame = r.integers(0, 256, (270, 480, 3), dtype=np.uint8)
    boxes = np.array([[240, 135, 200, 200], [10, 20, 120, 90], [470, 260, 300, 340], [100.5, 77.25, 33.3, 51.7]], np.float32)
    for scale in (1.0, 1.2):
        ref = crop_normalize_np(frame, boxes, scale)
        out = crop_normalize(torch.from_numpy(frame).to(cuda), torch.from_numpy(boxes).to(cuda), scale).cpu().numpy()
        # identical formula; fp32 rounding of the sample position may flip a .5 rounding of a grey level
        diff = np.abs(out - ref)
        assert (diff > 1e-5).mean() < 2e-3 and diff.max() <= 1.01 / 255 / 0.224


def test_demo_folder_end_to_end(tmp_path, cuda):
    from PIL import Image
    import demo
    from oracle import poco_ref
    from oracle.crop_np import crop_normalize_np
    variant = "resnet50-cliff"
    # synthetic checkpoint in the reference's format: {'state_dict': {'model.<part>.<key>': tensor}}
    w = util.synth_weights(variant)
    sd = {"model." + k: torch.from_numpy(v) for k, v in w.items()}
    sd["model.backbone.bn1.num_batches_tracked"] = torch.tensor(0)
    ckpt = tmp_path / "poco_synth.pt"
    torch.save({"state_dict": sd}, ckpt)
    smpl = synth.synth_smpl(7)
    np.savez(tmp_path / "smpl.npz", **smpl)
    imgs = tmp_path / "imgs"
    imgs.mkdir()
    r = np.random.default_rng(0)
    frames = {f"im{i}.png": r.integers(0, 256, (240, 320, 3), dtype=np.uint8) for i in range(2)}
    for n, f in frames.items():
        Image.fromarray(f).save(imgs / n)
    dets = {"im0.png": [[160, 120, 150, 150], [80, 100, 90, 120]], "im1.png": [[200, 100, 120, 160]]}
    (tmp_path / "dets.json").write_text(json.dumps(dets))
    args = demo.parse_args(["--cfg", "configs/demo_poco_cliff_resnet50.yaml", "--ckpt", str(ckpt), "--mode", "folder",
                            "--image_folder", str(imgs), "--output_folder", str(tmp_path / "out"), "--batch_size", "4",
                            "--smpl", str(tmp_path / "smpl.npz"), "--detections", str(tmp_path / "dets.json"), "--no_render"])
    demo.main(args)
    sd_t = poco_ref.to_torch(w)
    smpl_t = poco_ref.to_torch(smpl)
    for n, f in frames.items():
        res = dict(np.load(tmp_path / "out" / "imgs_" / (n[:-4] + "_poco.npz")))
        d = np.asarray(dets[n], np.float32)
        from poco_amd.tester import calculate_bbox_info, calculate_focal_length
        scale = np.maximum(d[:, 2], d[:, 3]) / 200.0
        batch = {"img": crop_normalize_np(f, d), "bbox_info": np.stack([calculate_bbox_info(c, s, (240, 320)) for c, s in zip(d[:, :2], scale)]),
                 "focal_length": np.full(len(d), calculate_focal_length(240, 320), np.float32), "scale": scale.astype(np.float32),
                 "center": d[:, :2].copy(), "orig_shape": np.tile([[240.0, 320.0]], (len(d), 1)).astype(np.float32)}
        ref = poco_ref.poco_forward(variant, sd_t, smpl_t, poco_ref.to_torch(batch))
        assert np.abs(res["pose"] - ref["pred_pose"].numpy()).max() < 1e-3
        assert np.abs(res["betas"] - ref["pred_shape"].numpy()).max() < 1e-3
        assert np.abs(res["verts"] - ref["smpl_vertices"].numpy()).max() < 1e-3
        assert res["var_global"].shape == (len(d),) and res["var_global"].max() <= 0.99
        assert res["smpl_joints2d"].shape == (len(d), 49, 3)
    # the reference's detector cache (joblib list indexed by image position, demo.py:163-169) gives the same results
    import joblib
    joblib.dump([np.asarray(dets[n], np.float32) for n in sorted(frames)], tmp_path / "detection_results.pkl")
    args.detections = str(tmp_path / "detection_results.pkl")
    args.output_folder = str(tmp_path / "out_pkl")
    demo.main(args)
    for n in frames:
        a = dict(np.load(tmp_path / "out" / "imgs_" / (n[:-4] + "_poco.npz")))
        b = dict(np.load(tmp_path / "out_pkl" / "imgs_" / (n[:-4] + "_poco.npz")))
        assert all(np.array_equal(a[k], b[k]) for k in a)
    # --save_obj: meshes/<image>/<idx>.obj as tester.py:300-303 (faces come from the body-model file when it has them)
    smpl_f = dict(smpl)
    smpl_f["faces"] = np.stack([np.arange(0, 300), np.arange(1, 301), np.arange(2, 302)], 1).astype(np.int32)
    np.savez(tmp_path / "smpl.npz", **smpl_f)
    args.save_obj = True
    args.output_folder = str(tmp_path / "out_obj")
    demo.main(args)
    res = dict(np.load(tmp_path / "out_obj" / "imgs_" / "im0_poco.npz"))
    lines = (tmp_path / "out_obj" / "imgs_" / "meshes" / "im0" / "000001.obj").read_text().splitlines()
    v = np.array([[float(t) for t in ln.split()[1:]] for ln in lines if ln.startswith("v ")])
    assert v.shape == (6890, 3) and np.abs(v - res["verts"][1]).max() < 1e-5
    assert sum(ln.startswith("f ") for ln in lines) == 300 and lines[6890] == "f 1 2 3"


def _tester(tmp_path, variant="resnet50-cliff", cfg="configs/demo_poco_cliff_resnet50.yaml", extra=()):
    import demo
    from poco_amd.tester import POCOTester
    w = util.synth_weights(variant)
    ckpt = tmp_path / "poco_synth.pt"
    torch.save({"state_dict": {"model." + k: torch.from_numpy(v) for k, v in w.items()}}, ckpt)
    np.savez(tmp_path / "smpl.npz", **synth.synth_smpl(7))
    args = demo.parse_args(["--cfg", cfg, "--ckpt", str(ckpt), "--mode", "video", "--vid_file", str(tmp_path),
                            "--output_folder", str(tmp_path / "out"), "--batch_size", "5",
                            "--smpl", str(tmp_path / "smpl.npz"), "--no_render", *extra])
    return POCOTester(args), args


def test_video_mode_tracks_match_frame_mode(tmp_path, cuda):
    """run_on_video (frame-major packing across frames/people, ragged last batch, a person entering late) returns,
    per track, exactly what the per-frame path regresses for the same boxes (tester.py:362-479 vs :153-245)."""
    from poco_amd import postproc
    t, _ = _tester(tmp_path)
    r = np.random.default_rng(5)
    frames = [r.integers(0, 256, (240, 320, 3), dtype=np.uint8) for _ in range(6)]
    tracks = {
        "a": {"bbox": np.array([[160 + 3 * i, 120, 150, 150] for i in range(6)], np.float32), "frames": np.arange(6)},
        "b": {"bbox": np.array([[80, 100 + 2 * i, 90, 120] for i in range(4)], np.float32), "frames": np.arange(2, 6)},
        "c": {"bbox": np.array([[250, 60, 70, 70]], np.float32), "frames": np.array([3])},
    }
    res = t.run_on_video(tracks, frames, 320, 240)
    assert set(res) == {"a", "b", "c"}
    for pid, tr in tracks.items():
        per_frame = t.run_on_frames([frames[f] for f in tr["frames"]], [tr["bbox"][k:k + 1] for k in range(len(tr["frames"]))])
        pose = np.concatenate([p["pose"] for p in per_frame])
        verts = np.concatenate([p["verts"] for p in per_frame])
        # different batch compositions -> different tile configurations: equal to rounding, not bitwise
        assert np.abs(res[pid]["pose"] - pose).max() < 1e-5
        assert np.abs(res[pid]["verts"] - verts).max() < 1e-5
        assert res[pid]["verts"].shape == (len(tr["frames"]), 6890, 3)
        assert res[pid]["smpl_joints2d"].shape == (len(tr["frames"]), 49, 2)
        assert np.array_equal(res[pid]["frame_ids"], tr["frames"]) and res[pid]["joints2d"] is None
        assert np.allclose(res[pid]["orig_cam"], postproc.convert_crop_cam_to_orig_img(res[pid]["pred_cam"], tr["bbox"], 320, 240))
        assert res[pid]["var"].shape == (len(tr["frames"]), 24) and res[pid]["var_global"].shape == (len(tr["frames"]),)


def test_video_mode_smoothing(tmp_path, cuda):
    """--smooth: filtered pose = the pinned one-euro restatement; verts/joints = LBS of the filtered pose with each
    frame's betas (smooth_pose.py:43-67), checked against the float64 oracle."""
    from oracle import smooth_np
    t, _ = _tester(tmp_path, extra=("--smooth",))
    r = np.random.default_rng(6)
    frames = [r.integers(0, 256, (200, 300, 3), dtype=np.uint8) for _ in range(7)]
    tracks = {"p": {"bbox": np.array([[150 + 4 * i, 100, 120, 140] for i in range(7)], np.float32), "frames": np.arange(7)}}
    t.args.smooth = False
    raw = t.run_on_video(tracks, frames, 300, 200)["p"]
    t.args.smooth = True
    sm = t.run_on_video(tracks, frames, 300, 200)["p"]
    v64, pose_hat, j64 = smooth_np.smooth_pose(raw["pose"], raw["betas"], synth.synth_smpl(7), 0.004, 1.5)
    assert np.abs(sm["pose"] - pose_hat).max() < 1e-6
    assert np.array_equal(sm["pose"][0], raw["pose"][0]) and np.abs(sm["pose"][3] - raw["pose"][3]).max() > 1e-4
    assert np.abs(sm["verts"] - v64).max() < 1e-4 and np.abs(sm["smpl_joints3d"] - j64).max() < 1e-4


def test_demo_video_cli(tmp_path, cuda):
    from PIL import Image
    import demo
    fr = tmp_path / "frames"
    fr.mkdir()
    r = np.random.default_rng(1)
    for i in range(3):
        Image.fromarray(r.integers(0, 256, (120, 160, 3), dtype=np.uint8)).save(fr / f"{i:06d}.png")
    t, args = _tester(tmp_path)
    args.vid_file = str(fr)
    stats = t.run_on_video_folder(str(fr), None, str(tmp_path / "out"))
    assert stats["images"] == 3 and stats["crops"] == 3 and stats["tracks"] == 1
    z = np.load(tmp_path / "out" / "poco_results.npz")
    assert z["0/verts"].shape == (3, 6890, 3) and z["0/var_global"].shape == (3,)


def test_crop_stream_matches_batch_path(tmp_path, cuda):
    """poco_amd.stream.CropStream (pinned frame ring, GPU crops into the resident batch, graph replay, packed
    253-float record) == POCOTester.make_batch + model(...) on the same frames / boxes."""
    from poco_amd.stream import CropStream, REC
    t, _ = _tester(tmp_path)
    r = np.random.default_rng(9)
    frames = [r.integers(0, 256, (180, 240, 3), dtype=np.uint8) for _ in range(3)]
    boxes = [np.array([[120, 90, 100, 100], [60, 70, 50, 80]], np.float32), np.array([[200, 50, 70, 60]], np.float32),
             np.array([[100, 100, 150, 120], [30, 40, 40, 40]], np.float32)]
    cs = CropStream(t.model, (180, 240), batch=5, ring=4)
    for rep in range(2):                       # second pass re-uses ring slots and the captured graph
        groups = [(cs.upload(f), b) for f, b in zip(frames, boxes)]
        rec, n = cs.run(groups, rep & 1)
        torch.cuda.synchronize()
        assert n == 5 and rec.shape == (5, REC)
        rec = rec.numpy().copy()
        row = 0
        for f, b in zip(frames, boxes):
            out = t.model(t.make_batch(torch.from_numpy(f).to(cuda), b), want_segm=False)
            k = len(b)
            assert np.abs(rec[row:row + k, :216] - out["pred_pose"].reshape(k, 216).cpu().numpy()).max() < 1e-5
            assert np.abs(rec[row:row + k, 216:226] - out["pred_shape"].cpu().numpy()).max() < 1e-5
            assert np.abs(rec[row:row + k, 226:229] - out["pred_cam"].cpu().numpy()).max() < 1e-5
            assert np.abs(rec[row:row + k, 229:253] - out["var_pose"].cpu().numpy()).max() < 1e-5
            row += k
