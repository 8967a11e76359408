"""CPU (no GPU): the oracle restatement against the committed golden vectors that were produced by
the reference's own modules (oracle/gen_golden.py), plus invariants for the SMPL step whose
arithmetic lives in the un-vendored smplx (parity unpinned there)."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import poco_ref, smpl_np
from poco_amd import synth

GOLD = Path(__file__).parent / "golden"
VARIANTS = ["hrnet_w32-pare", "hrnet_w48_cls-cliff", "resnet50-cliff"]


def load_spec(variant):
    return [(n, tuple(s)) for n, s in json.loads((GOLD / f"spec_{variant}.json").read_text())]


@pytest.fixture(scope="module")
def ops():
    return dict(np.load(GOLD / "ops.npz"))


def test_ops_rot6d(ops):
    out = poco_ref.rot6d_to_rotmat(torch.from_numpy(ops["rot6d_in"])).numpy()
    assert np.abs(out - ops["rot6d_out"]).max() < 1e-6
    # SURVEY 8(a) a9 check vector: input 1..6 -> first column (0.169, 0.507, 0.845)
    np.testing.assert_allclose(out[0][:, 0], [0.169031, 0.507093, 0.845154], atol=1e-5)
    R = out.astype(np.float64)
    assert np.abs(R.transpose(0, 2, 1) @ R - np.eye(3)).max() < 1e-5
    assert np.abs(np.linalg.det(R) - 1).max() < 1e-5


def test_ops_keypoint_attention(ops):
    out = poco_ref.keypoint_attention(torch.from_numpy(ops["ka_feat"]), torch.from_numpy(ops["ka_heat"])).numpy()
    assert np.abs(out - ops["ka_out"]).max() < 1e-5


def test_ops_lc2d(ops):
    x, w = torch.from_numpy(ops["lc_x"]), torch.from_numpy(ops["lc_w"])
    out = torch.einsum("bcj,ocj->boj", x[..., 0], w[0, :, :, :, 0, 0]).numpy()
    assert np.abs(out - ops["lc_out"][..., 0]).max() < 1e-5


def test_ops_cameras(ops):
    cam, pts = torch.from_numpy(ops["cam_in"]), torch.from_numpy(ops["cam_pts"])
    t = poco_ref.weak_persp_to_persp(cam)
    assert np.abs(t.numpy() - ops["cam_t"]).max() < 1e-6
    assert np.abs(poco_ref.project(pts, t, 5000.0, 0.0, 0.0).numpy() - ops["cam_proj"]).max() < 2e-3
    b = poco_ref.to_torch(synth.synth_batch(5, 77))
    tf = poco_ref.full_img_cam(cam, b["scale"] * 200.0, b["center"], b["orig_shape"][:, 1], b["orig_shape"][:, 0],
                               b["focal_length"])
    assert np.abs(tf.numpy() - ops["cam_full_t"]).max() < 1e-4
    proj = poco_ref.project(pts, tf, b["focal_length"], b["orig_shape"][:, 1] / 2, b["orig_shape"][:, 0] / 2)
    assert np.abs(proj.numpy() - ops["cam_full_proj"]).max() < 2e-2   # pixels, values up to ~1e4


@pytest.mark.parametrize("L", [1, 3])
def test_ops_realnvp(ops, L):
    import torch.nn as nn  # noqa
    spec = []
    for i in range(2 * L):
        for net in "ts":
            for k, (o, i_) in zip((0, 2, 4), ((64, 521), (64, 64), (9, 64))):
                spec += [(f"flow_head.flow.{net}.{i}.{k}.weight", (o, i_)), (f"flow_head.flow.{net}.{i}.{k}.bias", (o,))]
    spec += [("flow_head.flow.mask", (2 * L, 9)), ("flow_head.cond_layer.weight", (512, 64)),
             ("flow_head.cond_layer.bias", (512,))]
    sd = poco_ref.to_torch(synth.synth_state_dict(spec, 5))
    x, c = torch.from_numpy(ops[f"nvp{L}_x"]), torch.from_numpy(ops[f"nvp{L}_c"])
    assert np.abs(poco_ref.realnvp_log_prob(sd, x, c).numpy() - ops[f"nvp{L}_logp"]).max() < 1e-4
    z = torch.from_numpy(ops[f"nvp{L}_z"])
    fwd = poco_ref.realnvp_forward(sd, z, c)
    assert np.abs(fwd.numpy() - ops[f"nvp{L}_fwd"]).max() < 1e-4
    back, _ = poco_ref.realnvp_backward(sd, fwd, c)          # bijection: backward(forward(z)) == z
    assert np.abs(back.numpy() - ops[f"nvp{L}_z"]).max() < 1e-4


def test_ops_uncert(ops):
    assert np.array_equal(poco_ref.kinematic_uncert(ops["uncert_var"], synth.SMPL_PARENTS), ops["uncert_kin"])
    skel = ops["smpl_skeleton"]
    parents = -np.ones(24, int)
    parents[skel[:, 1]] = skel[:, 0]
    assert np.array_equal(parents, synth.SMPL_PARENTS)
    g = poco_ref.global_uncert(ops["uncert_var"], "hrnet_w48_cls-cliff")
    assert g.max() <= 0.99 and g.shape == (6,)


@pytest.mark.parametrize("profile", ["default", "stress"])
@pytest.mark.parametrize("variant", VARIANTS)
def test_model_golden(variant, profile):
    """Full oracle forward (B=2) reproduces what the reference modules produced in the build container, for both
    synthetic weight profiles (stress: every BN gamma in [0.5,1.5])."""
    torch.set_num_threads(8)
    tag = "" if profile == "default" else "_stress"
    g = dict(np.load(GOLD / f"model_{variant}{tag}.npz"))
    calib = synth.load_calib(variant) if profile == "stress" else None
    w = synth.synth_state_dict(load_spec(variant), 0, profile, calib)
    sd = poco_ref.to_torch({k: v for k, v in w.items() if v.dtype != np.int64})
    out = poco_ref.poco_forward(variant, sd, poco_ref.to_torch(synth.synth_smpl(7)),
                                poco_ref.to_torch(synth.synth_batch(2, 1234, profile=profile)))
    if profile == "stress":      # the fixture must not be vacuous: the two crops differ by >= 10x the 1e-3 gate
        for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose"):
            assert np.abs(g[k][0] - g[k][1]).max() >= 1e-2, k
    for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose"):
        assert np.abs(out[k].numpy() - g[k]).max() < 5e-5, k
    assert np.abs(out["uncert_feat"].numpy()[:, g["uncert_feat_idx"]] - g["uncert_feat_samples"]).max() < 2e-4
    assert np.abs(out["smpl_vertices"].numpy()[:, g["oracle_vert_idx"]] - g["oracle_vert_samples"]).max() < 1e-4
    assert np.abs(out["smpl_joints3d"].numpy() - g["oracle_smpl_joints3d"]).max() < 1e-4


def test_stress_profile_is_undamped():
    """BN gamma in [0.5,1.5] on EVERY BatchNorm incl. the last one of each residual branch and the
    fuse layers (the default profile damps those to ~0.1, which hid image-driven signal: VERDICT r1 weak #1)."""
    for variant in VARIANTS:
        spec = load_spec(variant)
        w = synth.synth_state_dict(spec, 0, "stress", synth.load_calib(variant))
        names = {n for n, _ in spec}
        gammas = [w[n] for n, _ in spec if n.endswith(".weight") and (n[:-7] + ".running_mean") in names]
        assert len(gammas) == len(synth.load_calib(variant))
        assert min(g.min() for g in gammas) >= 0.5 and max(g.max() for g in gammas) <= 1.5
        assert np.mean([g.mean() for g in gammas]) > 0.95


def test_smpl_invariants():
    smpl = synth.synth_smpl(7)
    st = poco_ref.to_torch(smpl)
    r = np.random.default_rng(3)
    eye = np.tile(np.eye(3, dtype=np.float32), (2, 24, 1, 1))
    zero = np.zeros((2, 10), np.float32)
    # identity pose + zero betas -> template
    v, _ = smpl_np.smpl_lbs_np(smpl, zero, eye)
    assert np.abs(v - smpl["v_template"][None]).max() < 1e-6
    # fp32 torch restatement == fp64 numpy restatement on random inputs
    betas = r.standard_normal((2, 10)).astype(np.float32)
    R = poco_ref.rot6d_to_rotmat(torch.from_numpy(r.standard_normal((48, 6)).astype(np.float32))).reshape(2, 24, 3, 3)
    v64, j64 = smpl_np.smpl_lbs_np(smpl, betas, R.numpy())
    v32, j32 = poco_ref.smpl_lbs(st, torch.from_numpy(betas), R)
    assert np.abs(v32.numpy() - v64).max() < 2e-5 and np.abs(j32.numpy() - j64).max() < 2e-5
    # rigid global rotation of the root rotates the whole mesh about the root joint
    Rg = R.numpy().copy()
    Q = poco_ref.rot6d_to_rotmat(torch.from_numpy(r.standard_normal((1, 6)).astype(np.float32))).numpy()[0]
    Rg2 = Rg.copy()
    Rg2[:, 0] = Q @ Rg[:, 0]
    va, ja = smpl_np.smpl_lbs_np(smpl, betas, Rg)
    vb, jb = smpl_np.smpl_lbs_np(smpl, betas, Rg2)
    root_a = ja[:, 8:9] * 0 + smpl_np.smpl_lbs_np(smpl, betas, Rg)[1][:, 8:9]  # joint_map[8] == 0 (pelvis)
    rel_a, rel_b = va - root_a, vb - root_a
    assert np.abs(rel_b - rel_a @ Q.T.astype(np.float64)).max() < 1e-5


# ---- crop: the cv2 fixed-point restatement (oracle/crop_np.py) -------------------------------------------------------------
def test_crop_points_pinned_to_reference(ops):
    """the float32 point triples handed to cv2.getAffineTransform were recorded from the reference's own
    gen_trans_from_patch_cv (vibe_image_utils.py:58-92, cv2 stubbed by a recorder; oracle/gen_golden.py)."""
    from oracle import crop_np
    for bx, sc, s, d in zip(ops["crop_boxes"], ops["crop_scale"], ops["crop_src"], ops["crop_dst"]):
        ps, pd = crop_np.patch_points(*bx, 224, sc)
        assert ps.dtype == np.float32 and np.array_equal(ps, s) and np.array_equal(pd, d)


def test_crop_bilinear_table_and_closed_form():
    """BilinearTab_i as initInterTab2D builds it: every entry sums to 2^15, entry (0,0) is {32767,0,0,1}, and the closed
    form the HIP kernel uses ((32-fy)(32-fx)*32, ... with that one exception) IS the table."""
    from oracle import crop_np
    tab = crop_np.bilinear_tab_i()
    assert tab.shape == (1024, 4) and (tab.sum(1) == 32768).all() and tab.min() >= 0 and tab.max() <= 32767
    assert tab[0].tolist() == [32767, 0, 0, 1]
    for fy in range(32):
        for fx in range(32):
            w = [(32 - fy) * (32 - fx) * 32, (32 - fy) * fx * 32, fy * (32 - fx) * 32, fy * fx * 32]
            if fx == 0 and fy == 0:
                w = [32767, 0, 0, 1]
            assert tab[fy * 32 + fx].tolist() == w


def test_crop_lu_solve_matches_exact_solution():
    """LUImpl<double> restated: on the affine system of an axis-aligned box the solution is the closed form
    [[s,0,tx],[0,s,ty]] up to double rounding; pivoting is exercised by a rotated triple."""
    from oracle import crop_np
    src, dst = crop_np.patch_points(640.25, 360.5, 300.0, 300.0, 224, 1.1)
    M = crop_np.get_affine_transform_cv(src, dst)
    s = 112.0 / float(src[2, 0] - src[0, 0])
    assert np.allclose(M, [[s, 0, 112 - s * float(src[0, 0])], [0, 112.0 / float(src[1, 1] - src[0, 1]), 112 - 112.0 / float(src[1, 1] - src[0, 1]) * float(src[0, 1])]],
                       rtol=1e-13, atol=1e-10)
    rot = np.float32([[0, 0], [0, 1], [1, 0]]) @ np.float32([[0.6, 0.8], [-0.8, 0.6]]) * 50 + np.float32([10, 20])
    M2 = crop_np.get_affine_transform_cv(rot.astype(np.float32), dst)
    back = (M2[:, :2] @ rot.astype(np.float64).T).T + M2[:, 2]
    assert np.abs(back - dst).max() < 1e-9
    assert np.abs(np.linalg.solve(np.array([[*p, 1.0] for p in rot], np.float64), dst.astype(np.float64)).T - M2).max() < 1e-10


def test_crop_fixed_point_properties():
    """identity box reproduces the image bytes; integer translations are exact copies; the result never differs from the
    exact-weight bilinear value by more than the 1/32-px quantisation allows; outside pixels are 0 (BORDER_CONSTANT)."""
    from oracle import crop_np
    r = np.random.default_rng(0)
    img = r.integers(0, 256, (300, 400, 3), dtype=np.uint8)
    assert np.array_equal(crop_np.crop_u8_np(img, [[112, 112, 224, 224]])[0], img[:224, :224])
    assert np.array_equal(crop_np.crop_u8_np(img, [[112 + 37, 112 + 21, 224, 224]])[0], img[21:245, 37:261])
    far = crop_np.crop_u8_np(img, [[-500, -500, 100, 100]])[0]
    assert far.max() == 0
    edge = crop_np.crop_u8_np(img, [[0, 0, 224, 224]])[0]            # top-left quadrant lies outside
    assert edge[:111, :111].max() == 0 and np.array_equal(edge[112:, 112:], img[:112, :112])
    yy, xx = np.mgrid[0:300, 0:400]
    smooth = np.stack([xx * 0.6 + 5, yy * 0.8 + 3, (xx + yy) * 0.35], -1).astype(np.uint8)
    boxes = np.array([[200.3, 150.7, 180.2, 160.9], [120, 90, 400, 333]], np.float64)
    u = crop_np.crop_u8_np(smooth, boxes, 1.1).astype(np.float64)
    f = crop_np.crop_normalize_float_np(smooth, boxes, 1.1)
    inner = (slice(None), slice(40, 180), slice(40, 180))
    assert np.abs(u - f)[inner].max() <= 1.6           # gradients < 1 grey level per px: 1/64 px + rounding
    # normalisation is ToTensor + Normalize in float32
    n = crop_np.normalize_np(np.full((1, 2, 2, 3), 255, np.uint8))
    assert np.allclose(n[0, :, 0, 0], (1 - crop_np.MEAN) / crop_np.STD, rtol=1e-6)
