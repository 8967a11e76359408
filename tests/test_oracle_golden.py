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
