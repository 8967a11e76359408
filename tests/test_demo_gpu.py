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
