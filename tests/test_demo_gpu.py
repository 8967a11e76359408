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
    """poco_crop_normalize == the cv2 restatement (oracle/crop_np.py: getAffineTransform's LU in double, warpAffine's 10-bit
    fixed-point coordinates, 1/32-px int16 weights, rounded 15-bit shift) BITWISE: the normalised float32 crops are compared
    with array_equal, for float32 and float64 boxes, boxes hanging over every image border, sub-pixel centres, tiny and
    huge boxes, several bbox scales."""
    from oracle.crop_np import crop_normalize_np
    from poco_amd.tester import crop_normalize
    r = np.random.default_rng(3)
    frame = r.integers(0, 256, (270, 480, 3), dtype=np.uint8)
    fixed = np.array([[240, 135, 200, 200], [10, 20, 120, 90], [470, 260, 300, 340], [100.5, 77.25, 33.3, 51.7],
                      [112, 112, 224, 224], [0, 0, 7, 9], [479, 269, 1500, 1200], [-50, -40, 60, 60]], np.float64)
    rand = np.stack([r.uniform(-40, 520, 40), r.uniform(-40, 310, 40), r.uniform(8, 700, 40), r.uniform(8, 700, 40)], 1)
    fd = torch.from_numpy(frame).to(cuda)
    n = 0
    for dt in (np.float32, np.float64):
        boxes = np.concatenate([fixed, rand]).astype(dt)
        for scale in (1.0, 1.1, 1.2):
            ref = crop_normalize_np(frame, boxes, scale)
            out = crop_normalize(fd, torch.from_numpy(boxes).to(cuda), scale).cpu().numpy()
            bad = np.argwhere(out != ref)
            assert bad.size == 0, (dt, scale, len(bad), bad[:4], boxes[bad[0][0]])
            n += len(boxes)
    print(f"{n} crops bit-identical to the fixed-point cv2 restatement")


def test_crop_normalize_full_hd_many_boxes(cuda):
    """BASELINE config #5 shape: 128 boxes on a 1080p frame, bitwise against the oracle."""
    from oracle.crop_np import crop_normalize_np
    from poco_amd.tester import crop_normalize
    r = np.random.default_rng(31)
    frame = r.integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    side = r.uniform(150, 600, 128)
    boxes = np.stack([r.uniform(0.1, 0.9, 128) * 1920, r.uniform(0.1, 0.9, 128) * 1080, side, side], 1).astype(np.float32)
    ref = crop_normalize_np(frame, boxes, 1.0)
    out = crop_normalize(torch.from_numpy(frame).to(cuda), torch.from_numpy(boxes).to(cuda), 1.0).cpu().numpy()
    assert np.array_equal(out, ref)


def test_crop_normalize_multi_frame_launch(cuda):
    """poco_crop_normalize_multi (one crop launch per streaming batch): crops cut from several frames through
    a device table of frame pointers + a per-crop frame index are BITWISE what poco_crop_normalize gives per frame (itself bitwise
    the cv2 restatement), in any crop order, frames used zero / one / many times."""
    import ctypes as C
    from poco_amd._lib import check, lib
    from poco_amd.tester import crop_normalize
    r = np.random.default_rng(21)
    H, W = 270, 480
    frames = [torch.from_numpy(r.integers(0, 256, (H, W, 3), dtype=np.uint8)).to(cuda) for _ in range(5)]
    N = 23
    fidx = r.integers(0, 4, N).astype(np.int32)                      # frame 4 is never used
    side = r.uniform(40, 400, N)
    boxes = np.stack([r.uniform(-0.1, 1.1, N) * W, r.uniform(-0.1, 1.1, N) * H, side, side * r.uniform(0.5, 1.5, N)], 1).astype(np.float32)
    ptrs = torch.tensor([f.data_ptr() for f in frames], dtype=torch.int64, device=cuda)
    bd, fd = torch.from_numpy(boxes).to(cuda), torch.from_numpy(fidx).to(cuda)
    out = torch.empty(N, 3, 224, 224, device=cuda)
    L = lib()
    L.poco_crop_normalize_multi.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_double,
                                            C.c_int, C.c_void_p, C.c_void_p]
    check(L.poco_crop_normalize_multi(ptrs.data_ptr(), 5, fd.data_ptr(), H, W, bd.data_ptr(), N, 1.1, 224, out.data_ptr(), None), "multi")
    torch.cuda.synchronize()
    for n in range(N):
        one = crop_normalize(frames[fidx[n]], bd[n:n + 1], 1.1)
        assert torch.equal(one[0], out[n]), n
    assert L.poco_crop_normalize_multi(None, 5, fd.data_ptr(), H, W, bd.data_ptr(), N, 1.1, 224, out.data_ptr(), None) != 0


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


def test_folder_mode_batches_across_images(tmp_path, cuda):
    """consecutive images share forwards (iter_frame_results) - frames of different sizes, a frame
    without detections in the middle, a frame with more people than the batch, float64 detections - and every image gets
    exactly the rows a one-forward-per-image run (the reference's loop, tester.py:168-213) gives it."""
    t, _ = _tester(tmp_path)                       # batch_size 5
    r = np.random.default_rng(21)
    sizes = [(240, 320), (200, 300), (240, 320), (180, 260), (240, 320), (300, 200), (240, 320)]
    frames = [r.integers(0, 256, (h, w, 3), dtype=np.uint8) for h, w in sizes]
    npeople = [1, 2, 0, 1, 7, 1, 3]
    dets = [np.stack([r.uniform(60, 200, n), r.uniform(50, 150, n), r.uniform(60, 160, n), r.uniform(60, 160, n)], 1).astype(
        np.float64 if i % 2 else np.float32) if n else np.zeros((0, 4), np.float32) for i, n in enumerate(npeople)]
    shared = t.run_on_frames(frames, dets)
    assert [None if x is None else len(x["pose"]) for x in shared] == [1, 2, None, 1, 7, 1, 3]
    one_by_one = [t.run_on_frames([f], [d])[0] for f, d in zip(frames, dets)]
    for a, b, n in zip(shared, one_by_one, npeople):
        if n == 0:
            assert a is None and b is None
            continue
        assert set(a) == set(b)
        for k in a:
            assert a[k].shape == b[k].shape, k
            # different batch compositions -> different tile configurations: equal to rounding, not bitwise
            assert np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max() < (1e-2 if k == "smpl_joints2d" else 2e-5), k
    # forwards: 15 crops at batch size 5 -> 3 full forwards instead of 6 per-image ones (+1 split of the 7-person frame)
    n_fw = []
    orig = type(t.model).__call__

    def counting(self, batch, *a, **k):
        n_fw.append(int(batch["img"].shape[0]))
        return orig(self, batch, *a, **k)

    type(t.model).__call__ = counting
    try:
        t.run_on_frames(frames, dets)
    finally:
        type(t.model).__call__ = orig
    assert n_fw == [5, 5, 5], n_fw


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
            assert np.abs(rec[row:row + k, 253] - out["record"][:, 253].cpu().numpy()).max() < 1e-5 and (rec[row:row + k, 253] <= 0.99).all()
            row += k
    # upload_many (staging copies on a thread pool) == upload frame by frame, bitwise; a frame without boxes frees its slot;
    # zero-copy staging through pinned_frame(slot)
    slots = cs.upload_many(frames + [frames[0]], threads=3)
    rec2, n2 = cs.run(list(zip(slots[:3], boxes)) + [(slots[3], np.zeros((0, 4), np.float32))], 0)
    torch.cuda.synchronize()
    assert n2 == 5 and np.array_equal(rec2.numpy()[:5], rec[:5]) and not any(cs._pending)
    assert cs.pinned_frame(slots[1]).shape == (180, 240, 3) and np.array_equal(cs.pinned_frame(slots[1]), frames[1])


def test_demo_folder_pare_b1_stress(tmp_path, cuda):
    """BASELINE.json config #1: POCO-PARE, ONE 224x224 crop, bs=1, through `demo.py --cfg configs/demo_poco_pare.yaml
    --mode folder`, against the CPU oracle on the very same crop.  Stress weights (undamped residual branches), so the
    1e-3 gate is far below the image-driven signal."""
    from PIL import Image
    import demo
    from oracle import poco_ref
    from oracle.crop_np import crop_normalize_np
    variant = "hrnet_w32-pare"
    w = util.synth_weights(variant, profile="stress")
    torch.save({"state_dict": {"model." + k: torch.from_numpy(v) for k, v in w.items()}}, tmp_path / "poco_pare_synth.pt")
    smpl = synth.synth_smpl(7)
    np.savez(tmp_path / "smpl.npz", **smpl)
    imgs = tmp_path / "one"
    imgs.mkdir()
    r = np.random.default_rng(11)
    # a 224x224 image with structure (blocks + noise): the default centred box is the whole image = "a 224x224 crop"
    frame = np.clip(128 + 60 * r.standard_normal((7, 7, 3)).repeat(32, 0).repeat(32, 1) + 25 * r.standard_normal((224, 224, 3)),
                    0, 255).astype(np.uint8)
    Image.fromarray(frame).save(imgs / "crop.png")
    args = demo.parse_args(["--cfg", "configs/demo_poco_pare.yaml", "--ckpt", str(tmp_path / "poco_pare_synth.pt"),
                            "--mode", "folder", "--image_folder", str(imgs), "--output_folder", str(tmp_path / "out"),
                            "--batch_size", "1", "--smpl", str(tmp_path / "smpl.npz"), "--no_render"])
    demo.main(args)
    res = dict(np.load(tmp_path / "out" / "one_" / "crop_poco.npz"))
    det = np.array([[112.0, 112.0, 224.0, 224.0]], np.float32)
    assert np.array_equal(res["bboxes"], det)
    batch = {"img": crop_normalize_np(frame, det)}
    ref = poco_ref.poco_forward(variant, poco_ref.to_torch(w), poco_ref.to_torch(smpl), poco_ref.to_torch(batch))
    dev = {"pose": np.abs(res["pose"] - ref["pred_pose"].numpy()).max(), "betas": np.abs(res["betas"] - ref["pred_shape"].numpy()).max(),
           "cam": np.abs(res["pred_cam"] - ref["pred_cam"].numpy()).max(), "verts": np.abs(res["verts"] - ref["smpl_vertices"].numpy()).max(),
           "joints3d": np.abs(res["joints3d"] - ref["smpl_joints3d"].numpy()).max()}
    from poco_amd import postproc
    var_ref, g_ref = postproc.folder_uncert(ref["var_pose"].numpy(), variant, True)
    dev["var"] = np.abs(res["var"] - var_ref).max()
    dev["var_global"] = np.abs(res["var_global"] - g_ref).max()
    print("PARE bs=1 via demo.py, max-abs deviation vs the CPU oracle:", {k: "%.2e" % v for k, v in dev.items()})
    assert max(dev.values()) < 1e-3, dev
    assert res["verts"].shape == (1, 6890, 3) and res["smpl_joints2d"].shape == (1, 49, 3)


def test_crop_stream_b128_w48_against_oracle(cuda):
    """BASELINE.json config #5 shape on one GPU: HRNet-W48-CLIFF at bs=128 through CropStream (pinned 1080p frame ring
    -> GPU crop/normalise -> hipGraph forward with the tuned table -> packed record), 4 people per frame; a 6-crop subset
    (first / middle / last rows) against the CPU oracle on numpy-cropped inputs.  Stress weights."""
    from oracle import poco_ref
    from oracle.crop_np import crop_normalize_np
    from poco_amd.stream import CropStream, REC
    from poco_amd.tester import calculate_bbox_info, calculate_focal_length
    variant, B, people, H, W = "hrnet_w48_cls-cliff", 128, 4, 1080, 1920
    m = util.make_engine(variant, max_batch=B, profile="stress")
    cs = CropStream(m, (H, W), B, ring=B // people)
    r = np.random.default_rng(21)
    base = [np.clip(128 + 70 * r.standard_normal((9, 16, 3)).repeat(120, 0).repeat(120, 1), 0, 255).astype(np.uint8) for _ in range(3)]
    frames = [np.clip(base[i % 3].astype(np.int16) + r.integers(-30, 30, (H, W, 3)), 0, 255).astype(np.uint8) for i in range(B // people)]
    boxes = [np.stack([np.array([r.uniform(0.2, 0.8) * W, r.uniform(0.3, 0.7) * H, s, s * r.uniform(0.8, 1.6)], np.float32)
                       for s in r.uniform(150, 500, people)]) for _ in frames]
    rec, n = cs.run([(cs.upload(f), b) for f, b in zip(frames, boxes)])
    torch.cuda.synchronize()
    assert n == B and rec.shape == (B, REC)
    rec = rec.numpy().copy()
    pick = [0, 1, 62, 65, 126, 127]
    crops, info, scl, ctr = [], [], [], []
    for row in pick:
        f, d = frames[row // people], boxes[row // people][row % people:row % people + 1]
        crops.append(crop_normalize_np(f, d)[0])
        s = float(max(d[0, 2], d[0, 3]) / 200.0)
        info.append(calculate_bbox_info(d[0, :2], s, (H, W))); scl.append(s); ctr.append(d[0, :2])
    batch = {"img": np.stack(crops), "bbox_info": np.stack(info), "focal_length": np.full(len(pick), calculate_focal_length(H, W), np.float32),
             "scale": np.array(scl, np.float32), "center": np.stack(ctr).astype(np.float32),
             "orig_shape": np.tile([[float(H), float(W)]], (len(pick), 1)).astype(np.float32)}
    torch.set_num_threads(16)
    ref = poco_ref.poco_forward(variant, poco_ref.to_torch(util.synth_weights(variant, profile="stress")),
                                poco_ref.to_torch(synth.synth_smpl(7)), poco_ref.to_torch(batch))
    got = rec[pick]
    dev = {"pose": np.abs(got[:, :216] - ref["pred_pose"].numpy().reshape(-1, 216)).max(),
           "betas": np.abs(got[:, 216:226] - ref["pred_shape"].numpy()).max(),
           "cam": np.abs(got[:, 226:229] - ref["pred_cam"].numpy()).max(),
           "var": np.abs(got[:, 229:253] - ref["var_pose"].numpy()).max()}
    spread = np.abs(ref["pred_pose"].numpy()[0] - ref["pred_pose"].numpy()[-1]).max()
    print("W48-CLIFF bs=128 streaming, max-abs deviation vs the CPU oracle:", {k: "%.2e" % v for k, v in dev.items()},
          "inter-crop spread of pose %.2e" % spread)
    # (the GPU crops are bit-identical to the numpy crops: test_crop_normalize_kernel)
    assert max(dev.values()) < 1e-3, dev
    assert spread > 1e-2


def test_crop_stream_pipelined_batches_keep_their_own_boxes(tmp_path, cuda):
    """two runs in flight with DIFFERENT boxes - the second run's host-side staging must not
    overwrite the first run's boxes / bbox_info before its H2D copy has executed.  Also the ring guard."""
    from poco_amd.stream import CropStream
    t, _ = _tester(tmp_path)
    r = np.random.default_rng(12)
    frames = [r.integers(0, 256, (180, 240, 3), dtype=np.uint8) for _ in range(4)]
    boxes_a = [np.array([[120, 90, 100, 100], [60, 70, 50, 80]], np.float32), np.array([[200, 50, 70, 60]], np.float32)]
    boxes_b = [np.array([[40, 120, 60, 90]], np.float32), np.array([[150, 100, 150, 120], [30, 40, 40, 40]], np.float32)]
    cs = CropStream(t.model, (180, 240), batch=5, ring=4)
    # reference results, one run at a time
    want = []
    for fr, bx in ((frames[:2], boxes_a), (frames[2:], boxes_b)):
        rec, n = cs.run([(cs.upload(f), b) for f, b in zip(fr, bx)], 0)
        torch.cuda.synchronize()
        want.append(rec.numpy()[:n].copy())
    assert np.abs(want[0] - want[1]).max() > 1e-3
    # now back to back without a host sync in between; a long-running kernel keeps the stream busy so that the second
    # run()'s host code executes while the first run's H2D copy is still queued
    big = torch.randn(8192, 8192, device=cuda)
    for _ in range(3):
        for _ in range(6):
            big @ big
        ra, na = cs.run([(cs.upload(f), b) for f, b in zip(frames[:2], boxes_a)], 0)
        rb, nb = cs.run([(cs.upload(f), b) for f, b in zip(frames[2:], boxes_b)], 1)
        torch.cuda.synchronize()
        assert np.array_equal(ra.numpy()[:na], want[0]) and np.array_equal(rb.numpy()[:nb], want[1])
    # ring guard: a third upload round without run() would overwrite unconsumed frames
    cs2 = CropStream(t.model, (180, 240), batch=5, ring=2)
    cs2.upload(frames[0]); cs2.upload(frames[1])
    with pytest.raises(RuntimeError, match="ring"):
        cs2.upload(frames[2])
    # upload_many on a ring that cannot take all frames: refused BEFORE any slot is reserved (an overflow used to
    # shrink the ring for ever), so the ring works again as soon as run() / release() frees the busy slot
    cs3 = CropStream(t.model, (180, 240), batch=5, ring=3)
    k0 = cs3.upload(frames[0])
    with pytest.raises(RuntimeError, match="ring"):
        cs3.upload_many(frames[:3])
    assert cs3._pending == [True, False, False]
    cs3.release(k0)
    slots = cs3.upload_many(frames[:3])
    assert sorted(slots) == [0, 1, 2]
    rec, n = cs3.run([(slots[0], boxes_a[0]), (slots[1], boxes_a[1]), (slots[2], np.zeros((0, 4), np.float32))], 0)
    torch.cuda.synchronize()
    assert n == 3 and np.array_equal(rec.numpy()[:3], want[0]) and cs3._pending == [False] * 3
