"""CPU: tools/validate_assets.py (the opt-in path to pin SMPL-LBS / real checkpoints for users who hold the licences) and
tools/convert_smpl.py, exercised on SYNTHETIC assets in the reference's file formats; plus the opt-in test that compares the LBS
oracle with the real smplx layer when smplx and a model directory are present (skipped in this image)."""
import os
import pickle
import sys
from pathlib import Path

import numpy as np
import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from poco_amd import synth  # noqa: E402
from tests import util  # noqa: E402
from tools import validate_assets as va  # noqa: E402


def _fake_smpl_pickle(path, smpl):
    """A pickle with the keys / layouts of SMPL_NEUTRAL.pkl (scipy-sparse J_regressor, posedirs [V,3,207], kintree_table)."""
    import scipy.sparse as sp
    V = smpl["v_template"].shape[0]
    kin = np.stack([np.where(smpl["parents"] < 0, 2 ** 32 - 1, smpl["parents"]).astype(np.int64), np.arange(24)])
    d = {"v_template": smpl["v_template"].astype(np.float64), "shapedirs": np.concatenate([smpl["shapedirs"], np.zeros((V, 3, 290))], 2),
         "posedirs": smpl["posedirs"].T.reshape(V, 3, 207).astype(np.float64), "J_regressor": sp.csc_matrix(smpl["J_regressor"]),
         "weights": smpl["lbs_weights"].astype(np.float64), "kintree_table": kin, "f": np.zeros((10, 3), np.uint32)}
    with open(path, "wb") as f:
        pickle.dump(d, f, protocol=2)


def test_convert_and_validate_synthetic_smpl(tmp_path):
    smpl = synth.synth_smpl(7)
    _fake_smpl_pickle(tmp_path / "SMPL_NEUTRAL.pkl", smpl)
    np.save(tmp_path / "J_regressor_extra.npy", smpl["J_regressor_extra"])
    args = va.argparse.Namespace(smpl_pkl=str(tmp_path / "SMPL_NEUTRAL.pkl"), extra=str(tmp_path / "J_regressor_extra.npy"),
                                 smpl_npz=None, smpl_dir=None, ckpt=None, cfg=None, inf_model="best", device="cuda:0")
    st, msg, got = va.stage_smpl_file(args)
    assert st == va.OK, msg
    for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights", "J_regressor_extra"):
        assert np.allclose(got[k], smpl[k], atol=1e-7), k                       # the converter round-trips the pickle layout
    assert np.array_equal(got["parents"], smpl["parents"]) and got["faces"].shape == (10, 3)
    # structural problems are reported, not swallowed
    bad = dict(got); bad["lbs_weights"] = got["lbs_weights"] * 0.5
    assert any("sum to 1" in b for b in va.check_smpl_npz(bad))
    bad = dict(got); bad["posedirs"] = got["posedirs"].T
    assert any("posedirs" in b for b in va.check_smpl_npz(bad))
    # stage 2 without smplx: skipped, never failed
    st2, msg2 = va.stage_lbs(args, got)
    assert st2 == va.SKIP


def test_lbs_comparison_helper_detects_a_wrong_reference():
    smpl = synth.synth_smpl(7)
    from oracle import poco_ref
    st = poco_ref.to_torch(smpl)

    def good(betas, R):
        v, j = poco_ref.smpl_lbs(st, torch.from_numpy(betas), torch.from_numpy(R))
        return v.numpy(), j.numpy()

    dv, dj, *_ = va.lbs_against(good, smpl)
    assert dv < 2e-5 and dj < 2e-5
    dv, dj, *_ = va.lbs_against(lambda b, R: tuple(x + 1e-3 for x in good(b, R)), smpl)
    assert dv > 5e-4


def test_checkpoint_report_strict(tmp_path):
    from poco_amd.checkpoint import read_checkpoint
    from poco_amd.model import POCO
    variant = "resnet50-cliff"
    w = util.synth_weights(variant)
    sd = {"model." + k: torch.from_numpy(v) for k, v in w.items()}
    sd["model.backbone.bn1.num_batches_tracked"] = torch.tensor(0)
    sd["model.smpl.smpl.betas"] = torch.zeros(1, 10)                       # non-part keys of a Lightning checkpoint are ignored
    torch.save({"state_dict": sd}, tmp_path / "poco_synth.pt")
    eng = POCO(backbone=variant, num_flow_layers=1, max_batch=1)
    loaded = read_checkpoint(str(tmp_path / "poco_synth.pt"))
    rep = va.checkpoint_report(eng.expected_tensors(), loaded)
    assert rep["ok"] and not rep["missing"] and not rep["unexpected"], rep
    broken = dict(loaded)
    del broken["head.fc1.weight"]
    broken["head.decpose.weight"] = broken["head.decpose.weight"].T.copy()
    broken["backbone.layer9.conv.weight"] = np.zeros((1,), np.float32)
    rep = va.checkpoint_report(eng.expected_tensors(), broken)
    assert not rep["ok"] and rep["missing"] == ["head.fc1.weight"] and rep["unexpected"] == ["backbone.layer9.conv.weight"]
    assert len(rep["shape_mismatch"]) == 1 and "decpose" in rep["shape_mismatch"][0]
    # whole stage without GPU / SMPL: strict load passes, forward not run
    args = va.argparse.Namespace(smpl_pkl=None, extra=None, smpl_npz=None, smpl_dir=None, ckpt=str(tmp_path / "poco_synth.pt"),
                                 cfg="configs/demo_poco_cliff_resnet50.yaml", inf_model="best", device="cuda:0")
    if not torch.cuda.is_available():
        st, msg = va.stage_checkpoint(args, None)
        assert st == va.OK and "strict load OK" in msg, msg


def test_cli_skips_cleanly_without_assets(tmp_path, monkeypatch):
    monkeypatch.chdir(tmp_path)                                            # no data/ here
    assert va.main([]) == 0


@pytest.mark.skipif(not (os.environ.get("SMPL_MODEL_DIR") and os.path.isdir(os.environ.get("SMPL_MODEL_DIR", ""))),
                    reason="opt-in: set SMPL_MODEL_DIR (folder with SMPL_NEUTRAL.pkl + J_regressor_extra.npy) and install smplx==0.1.28")
def test_lbs_oracle_against_real_smplx():
    """Pins row a10 (smplx.lbs via pocolib/models/head/smpl_head.py:22-34) on a machine that has the licensed files."""
    pytest.importorskip("smplx")
    d = os.environ["SMPL_MODEL_DIR"]
    args = va.argparse.Namespace(smpl_pkl=os.path.join(d, "SMPL_NEUTRAL.pkl"), extra=os.path.join(d, "J_regressor_extra.npy"),
                                 smpl_npz=None, smpl_dir=d, ckpt=None, cfg=None, inf_model="best", device="cuda:0")
    st, msg, smpl = va.stage_smpl_file(args)
    assert st == va.OK, msg
    st, msg = va.stage_lbs(args, smpl)
    assert st == va.OK, msg


def test_crop_stage_logic_and_skip():
    """Stage 4 (crop vs the real cv2): skips cleanly where cv2 is absent (this image); its comparison logic is exercised with a
    stand-in `cv2` built from the oracle itself (must report zero differences) and with a deliberately different warp (exact
    float weights, what the round-2 kernel did: must be reported as differing)."""
    import types
    from oracle import crop_np
    try:
        import cv2  # noqa: F401
        have = True
    except Exception:
        have = False
    args = va.argparse.Namespace(device="cuda:0")
    if not have:
        st, msg = va.stage_crop(args)
        assert st == va.SKIP and "cv2" in msg
    fake = types.SimpleNamespace(INTER_LINEAR=1, BORDER_CONSTANT=0, __version__="stand-in",
                                 getAffineTransform=lambda s, d: crop_np.get_affine_transform_cv(s, d),
                                 warpAffine=lambda img, M, size, flags, borderMode: crop_np.warp_affine_u8(img, M, size[0]))
    w = va.crop_against_cv2(fake, n=3)
    assert w["matrix_bits"] == 0 and w["u8_pixels"] == 0

    def float_warp(img, M, size, flags, borderMode):
        Mi = crop_np.invert_affine_cv(M)
        ys, xs = np.meshgrid(np.arange(size[1], dtype=np.float64), np.arange(size[0], dtype=np.float64), indexing="ij")
        sx, sy = Mi[0] * xs + Mi[1] * ys + Mi[2], Mi[3] * xs + Mi[4] * ys + Mi[5]
        x0, y0 = np.floor(sx).astype(int), np.floor(sy).astype(int)
        fx, fy = (sx - x0)[..., None], (sy - y0)[..., None]
        H, W = img.shape[:2]

        def px(xx, yy):
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            return np.where(ok[..., None], img[np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)].astype(np.float64), 0.0)

        v = (1 - fy) * ((1 - fx) * px(x0, y0) + fx * px(x0 + 1, y0)) + fy * ((1 - fx) * px(x0, y0 + 1) + fx * px(x0 + 1, y0 + 1))
        return np.clip(np.rint(v), 0, 255).astype(np.uint8)

    fake.warpAffine = float_warp
    w2 = va.crop_against_cv2(fake, n=3)
    assert w2["u8_pixels"] > 0
