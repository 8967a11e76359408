"""ORACLE tooling: generate tests/golden/* by running the REFERENCE's own modules (imported from
/root/reference with stubbed third-party deps, CPU fp32) on seeded synthetic weights/inputs, and
pin oracle/poco_ref.py against them.

    python oracle/gen_golden.py            # run in the build container (needs /root/reference)

What is recorded (data only - inputs are re-derived from seeds by poco_amd/synth.py):
  tests/golden/spec_<variant>.json     state_dict key -> shape of the reference modules
  tests/golden/model_<variant>.npz     reference outputs for B=2 (pose/shape/cam/var/features ...)
  tests/golden/model_<variant>_stress.npz   the same with the "stress" weight profile (every BN gamma in [0.5,1.5])
  poco_amd/calib/stress_<variant>.json ONLY with the argument `calibrate`: the per-BN input statistics the
                                       stress profile needs (data; poco_amd/synth.py reads it on every machine)
  tests/golden/smooth.npz              One Euro filtered rotation tracks (one_euro_filter.py via smooth_pose.py)
  tests/golden/ops.npz                 per-op vectors (KeypointAttention, LocallyConnected2d,
                                       rot6d_to_rotmat, camera conversions, RealNVP, uncert post-proc)
The SMPL step cannot be run through the reference (smplx + SMPL assets absent): the vertices /
joints in the fixtures come from oracle/smpl_np.py (float64) and are labelled `oracle_*`.
"""
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import poco_ref, ref_import, smpl_np  # noqa: E402
from poco_amd import synth  # noqa: E402

GOLD = ROOT / "tests" / "golden"
VARIANTS = {
    # variant: POCO kwargs from configs/demo_poco_{pare,cliff}.yaml
    "hrnet_w32-pare": dict(num_neurons="512-", uncert_inp_type="feat-pose", num_flow_layers=3),
    "hrnet_w48_cls-cliff": dict(num_neurons="216-", uncert_inp_type="feat-pose-net", num_flow_layers=1),
    "resnet50-cliff": dict(num_neurons="216-", uncert_inp_type="feat-pose-net", num_flow_layers=1),
}
SEED_W, SEED_IN, BATCH = 0, 1234, 2


def build_reference(variant, kw):
    from pocolib.models.backbone.hrnet import hrnet_w32
    from pocolib.models.backbone.hrnet_cls import hrnet_w48_cls
    from pocolib.models.backbone.resnet import resnet50
    from pocolib.models.head import cliff_head, pare_head, poco_head
    from pocolib.models.head.nf_head import flow_head
    bname, hname = variant.split("-")
    backbone = {"hrnet_w32": hrnet_w32, "hrnet_w48_cls": hrnet_w48_cls, "resnet50": resnet50}[bname](pretrained=False)
    nch = {"hrnet_w32": 480, "hrnet_w48_cls": 2048, "resnet50": 2048}[bname]
    head = (cliff_head if hname == "cliff" else pare_head)(nch, "diff_branch", "sigmoid")
    num_neurons = list(map(int, filter(None, kw["num_neurons"].split("-"))))
    uncert = poco_head(head.get_output_channels(), num_neurons, 1, "sigmoid", True, "diff_branch", [],
                       "norm_flow_res_gaus", "pose", kw["uncert_inp_type"], False, "h36m", 0.25)
    flow = flow_head("pose", kw["num_flow_layers"], "", "alter", [], 9, True, head.get_output_channels(), 512)
    parts = {"backbone": backbone, "head": head, "uncert_head": uncert, "flow_head": flow}
    for m in parts.values():
        m.eval()
    return parts


def sample_idx(n, k, seed):
    return np.sort(np.random.default_rng(seed).choice(n, size=min(k, n), replace=False))


def ref_spec(parts):
    spec = []
    for pname, m in parts.items():
        for k, v in m.state_dict().items():
            spec.append((f"{pname}.{k}", tuple(v.shape)))
    return spec


def load_into(parts, w):
    for pname, m in parts.items():
        sd = {k[len(pname) + 1:]: torch.from_numpy(v) for k, v in w.items() if k.startswith(pname + ".")}
        m.load_state_dict(sd, strict=True)


def ref_forward(parts, variant, batch):
    hname = variant.split("-")[1]
    feats = parts["backbone"](batch["img"])
    return feats, (parts["head"](feats, batch) if hname == "cliff" else parts["head"](feats))


CALIB_BATCH, CALIB_SEED = 4, 4242


def calibrate(variant, kw):
    """"stress" weight profile (poco_amd/synth.py): measure, layer by layer in execution order, the mean and
    variance of every BatchNorm2d's input on a seeded batch through the REFERENCE modules, with all earlier BNs
    already carrying their calibrated running statistics.  Writes poco_amd/calib/stress_<variant>.json."""
    parts = build_reference(variant, kw)
    spec = ref_spec(parts)
    load_into(parts, synth.synth_state_dict(spec, SEED_W, "stress", {}))
    calib = {}

    def make_hook(full):
        def hook(mod, inp):
            x = inp[0]
            m, v = float(np.float32(x.mean())), float(np.float32(x.var()))
            calib[full] = [m, v]
            mini = [(f"{full}.{leaf}", tuple(mod.running_mean.shape)) for leaf in ("running_mean", "running_var")]
            t = synth.synth_state_dict(mini, SEED_W, "stress", {full: (m, v)})
            mod.running_mean.copy_(torch.from_numpy(t[f"{full}.running_mean"]))
            mod.running_var.copy_(torch.from_numpy(t[f"{full}.running_var"]))
        return hook

    for pname, m in parts.items():
        for name, mod in m.named_modules():
            if isinstance(mod, (torch.nn.BatchNorm2d, torch.nn.BatchNorm1d)):
                mod.register_forward_pre_hook(make_hook(f"{pname}.{name}"))
    with torch.no_grad():
        feats, ho = ref_forward(parts, variant, poco_ref.to_torch(synth.synth_batch(CALIB_BATCH, CALIB_SEED, profile="stress")))
    nbn = sum(1 for n, _ in spec if n.endswith(".running_mean"))
    assert len(calib) == nbn, (len(calib), nbn)
    synth.CALIB_DIR.mkdir(exist_ok=True)
    (synth.CALIB_DIR / f"stress_{variant}.json").write_text(json.dumps(calib, indent=0, sort_keys=True))
    vs = np.array([v for _, v in calib.values()])
    print(variant, f"calibrated {nbn} BNs; input variance range [{vs.min():.3g}, {vs.max():.3g}];",
          "feat rms", float(feats.pow(2).mean().sqrt()))


def run_variant(variant, kw, profile="default"):
    parts = build_reference(variant, kw)
    spec = ref_spec(parts)
    tag = "" if profile == "default" else f"_{profile}"
    if profile == "default":
        (GOLD / f"spec_{variant}.json").write_text(json.dumps([[n, list(s)] for n, s in spec]))
        w = synth.synth_state_dict(spec, SEED_W)
    else:
        w = synth.synth_state_dict(spec, SEED_W, profile, synth.load_calib(variant))
    load_into(parts, w)
    batch_np = synth.synth_batch(BATCH, SEED_IN, profile=profile)
    batch = poco_ref.to_torch(batch_np)
    smpl_np_model = synth.synth_smpl(7)
    smpl_t = poco_ref.to_torch(smpl_np_model)
    hname = variant.split("-")[1]
    with torch.no_grad():
        feats, ho = ref_forward(parts, variant, batch)
        verts64, j49_64 = smpl_np.smpl_lbs_np(smpl_np_model, ho["pred_shape"].numpy(), ho["pred_pose"].numpy())
        so = {"smpl_vertices": torch.from_numpy(verts64).float()}
        uo = parts["uncert_head"](ho, so, batch)
        fo = parts["flow_head"](ho, dict(so), batch)
        assert fo["log_phi"] is None
    # --- pin the oracle restatement against the reference modules -------------------------------
    sd_t = poco_ref.to_torch({k: v for k, v in w.items() if v.dtype != np.int64})
    mine = poco_ref.poco_forward(variant, sd_t, smpl_t, batch)
    pins = {}
    for key_ref, key_mine in [("pred_pose", "pred_pose"), ("pred_shape", "pred_shape"), ("pred_cam", "pred_cam"),
                              ("uncert_feat", "uncert_feat")]:
        pins[key_ref] = float((ho[key_ref] - mine[key_mine]).abs().max())
    pins["var_pose"] = float((uo["var_pose"] - mine["var_pose"]).abs().max())
    pins["smpl_vertices(fp32 torch vs f64 numpy)"] = float(np.abs(mine["smpl_vertices"].numpy() - verts64).max())
    if hname == "pare":
        pins["pred_segm_mask"] = float((ho["pred_segm_mask"] - mine["pred_segm_mask"]).abs().max())
    print(variant, profile, "oracle-vs-reference max abs:", json.dumps(pins))
    assert max(pins.values()) < 2e-4, pins
    # --- fixtures -----------------------------------------------------------------------------
    f_flat = feats.reshape(BATCH, -1).numpy()
    fidx = sample_idx(f_flat.shape[1], 512, 11)
    out = {
        "pred_pose": ho["pred_pose"].numpy(), "pred_shape": ho["pred_shape"].numpy(),
        "pred_cam": ho["pred_cam"].numpy(), "var_pose": uo["var_pose"].numpy(),
        "pred_pose6d": (ho["pred_pose6d"] if hname == "pare" else ho["pred_pose_6d"]).numpy().reshape(BATCH, -1),
        "feat_idx": fidx, "feat_samples": f_flat[:, fidx], "feat_sum": f_flat.astype(np.float64).sum(1),
        "feat_abs_mean": np.abs(f_flat).mean(1),
        "uncert_feat_idx": sample_idx(ho["uncert_feat"].shape[1], 256, 12),
        "oracle_smpl_joints3d": j49_64.astype(np.float32),
        "oracle_vert_idx": sample_idx(6890, 512, 13),
        "oracle_smpl_joints2d": mine["smpl_joints2d"].numpy(),
        "oracle_pred_cam_t": mine["pred_cam_t"].numpy(),
    }
    out["uncert_feat_samples"] = ho["uncert_feat"].numpy()[:, out["uncert_feat_idx"]]
    out["oracle_vert_samples"] = verts64[:, out["oracle_vert_idx"]].astype(np.float32)
    if hname == "cliff":
        out["body_feat2_samples"] = ho["body_feat2"].numpy()[:, :64]
        out["oracle_pred_fullimg_cam_t"] = mine["pred_fullimg_cam_t"].numpy()
    else:
        m = ho["pred_segm_mask"].reshape(BATCH, -1).numpy()
        out["segm_idx"] = sample_idx(m.shape[1], 256, 14)
        out["segm_samples"] = m[:, out["segm_idx"]]
    np.savez_compressed(GOLD / f"model_{variant}{tag}.npz", **out)
    spread = {k: float(np.abs(out[k][0] - out[k][1]).max()) for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose")}
    spread["vertices"] = float(np.abs(verts64[0] - verts64[1]).max())
    print(variant, profile, "inter-crop spread (max |crop0 - crop1|):", json.dumps(spread))
    print(variant, "feat |mean|", out["feat_abs_mean"], "var_pose range", float(uo["var_pose"].min()),
          float(uo["var_pose"].max()), "cam", ho["pred_cam"].numpy().round(3).tolist())


def run_ops():
    import importlib.util

    def load(path, name):
        spec = importlib.util.spec_from_file_location(name, os.path.join(ref_import.REFERENCE, path))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    geo = load("pocolib/utils/geometry.py", "ref_geometry")
    from pocolib.models.head.nf_head import flow_head
    from pocolib.models.head.smplcam_head import convert_pare_to_full_img_cam
    from pocolib.models.head.smplcam_head import perspective_projection as persp_K
    from pocolib.models.layers import KeypointAttention, LocallyConnected2d
    kp = load("pocolib/utils/kp_utils.py", "ref_kp_utils")
    r = np.random.default_rng(99)
    o = {}
    # KeypointAttention (keypoint_attention.py:34-48)
    feat = torch.from_numpy(r.standard_normal((2, 20, 9, 7)).astype(np.float32))
    heat = torch.from_numpy((3 * r.standard_normal((2, 24, 9, 7))).astype(np.float32))
    o["ka_feat"], o["ka_heat"] = feat.numpy(), heat.numpy()
    o["ka_out"] = KeypointAttention(use_conv=False, in_channels=(20, 8), out_channels=(20, 8))(feat, heat).numpy()
    # LocallyConnected2d (locallyconnected2d.py:27-37)
    lc = LocallyConnected2d(in_channels=128, out_channels=6, output_size=[24, 1], kernel_size=1, stride=1)
    with torch.no_grad():       # seeded weights: the module's own init draws from torch's global RNG (VERDICT r1 c)
        lc.weight.copy_(torch.from_numpy(synth.synth_state_dict([("lc.weight", tuple(lc.weight.shape))], 3)["lc.weight"]))
    x = torch.from_numpy(r.standard_normal((3, 128, 24, 1)).astype(np.float32))
    with torch.no_grad():
        o["lc_x"], o["lc_w"], o["lc_out"] = x.numpy(), lc.weight.detach().numpy(), lc(x).numpy()
    # rot6d_to_rotmat (geometry.py:247-261); first row is the SURVEY check vector 1..6
    x6 = np.concatenate([np.arange(1, 7, dtype=np.float32)[None], r.standard_normal((47, 6)).astype(np.float32)])
    o["rot6d_in"] = x6
    o["rot6d_out"] = geo.rot6d_to_rotmat(torch.from_numpy(x6)).numpy()
    # cameras
    cam = torch.from_numpy(np.stack([r.uniform(0.5, 1.2, 5), r.uniform(-.3, .3, 5), r.uniform(-.3, .3, 5)], 1).astype(np.float32))
    pts = torch.from_numpy(r.uniform(-1, 1, (5, 49, 3)).astype(np.float32))
    o["cam_in"], o["cam_pts"] = cam.numpy(), pts.numpy()
    t = geo.convert_weak_perspective_to_perspective(cam)
    o["cam_t"] = t.numpy()
    o["cam_proj"] = geo.perspective_projection(pts, torch.eye(3)[None].expand(5, -1, -1), t, 5000.0,
                                               torch.zeros(5, 2)).numpy()
    b = synth.synth_batch(5, 77)
    bt = poco_ref.to_torch(b)
    tf = convert_pare_to_full_img_cam(cam, bt["scale"] * 200.0, bt["center"], bt["orig_shape"][:, 1],
                                      bt["orig_shape"][:, 0], bt["focal_length"])
    K = torch.eye(3).repeat(5, 1, 1)
    K[:, 0, 0] = bt["focal_length"]; K[:, 1, 1] = bt["focal_length"]
    K[:, 0, 2] = bt["orig_shape"][:, 1] / 2.0; K[:, 1, 2] = bt["orig_shape"][:, 0] / 2.0
    o["cam_full_t"] = tf.numpy()
    o["cam_full_proj"] = persp_K(pts, torch.eye(3)[None].expand(5, -1, -1), tf, K).numpy()
    # RealNVP (real_nvp.py) via flow_head, both depths used by the shipped configs
    for L in (1, 3):
        fh = flow_head("pose", L, "", "alter", [], 9, True, 64, 512).eval()
        spec = [(f"flow_head.{k}", tuple(v.shape)) for k, v in fh.state_dict().items()]
        w = synth.synth_state_dict(spec, 5)
        fh.load_state_dict({k[len("flow_head."):]: torch.from_numpy(v) for k, v in w.items()}, strict=True)
        x = torch.from_numpy(np.abs(r.standard_normal((48, 9))).astype(np.float32))
        c = torch.from_numpy(r.standard_normal((48, 512)).astype(np.float32))
        with torch.no_grad():
            o[f"nvp{L}_x"], o[f"nvp{L}_c"] = x.numpy(), c.numpy()
            o[f"nvp{L}_logp"] = fh.flow.log_prob(x, c).numpy()
            z = torch.from_numpy(r.standard_normal((48, 9)).astype(np.float32))
            o[f"nvp{L}_z"] = z.numpy()
            o[f"nvp{L}_fwd"] = fh.flow.forward_p(z, c).numpy()
        sd_t = poco_ref.to_torch(w)
        assert float((poco_ref.realnvp_log_prob(sd_t, x, c) - torch.from_numpy(o[f"nvp{L}_logp"])).abs().max()) < 1e-4
        assert float((poco_ref.realnvp_forward(sd_t, z, c) - torch.from_numpy(o[f"nvp{L}_fwd"])).abs().max()) < 1e-4
    # uncertainty post-processing (poco_utils.py:21-25,50-60 restated through the reference skeleton table)
    skel = kp.get_smpl_skeleton()
    o["smpl_skeleton"] = skel
    var = r.uniform(0, 1, (6, 24)).astype(np.float32)
    v2 = var.copy()
    for i in skel[:, 1]:
        v2[:, i] += v2[:, skel[i - 1, 0]]
    o["uncert_var"], o["uncert_kin"] = var, v2
    # ---- host formulas, produced by the reference's own functions (a16 / a17 / a18) ---------------------------
    hu = ref_import.setup_host_utils()
    assert np.array_equal(hu["poco_utils"].get_kinematic_uncert(var.copy()), v2)
    var_hi = var.copy()
    var_hi[1, 0], var_hi[4, 0] = 0.93, 0.55          # rows above the cliff (0.8) / pare (0.4) thresholds
    o["uncert_var2"] = var_hi
    for bb in ("hrnet_w48_cls-cliff", "hrnet_w32-pare"):
        for kin in (True, False):
            pu = ref_import.poco_utils_instance(hu, bb, kin)
            tag = f"{bb.split('-')[1]}_{'kin' if kin else 'nokin'}"
            v = pu.prepare_uncert(torch.from_numpy(var_hi.copy()))                 # folder mode, tester.py:242-245
            o[f"uncert_{tag}_var"] = v.copy()
            o[f"uncert_{tag}_global"] = np.clip(pu.get_global_uncert(v.copy()), 0, 0.99)
            vt = pu.prepare_uncert(torch.from_numpy(var_hi.copy()), True)          # video mode, tester.py:416-419
            o[f"uncert_{tag}_var_video"] = vt.clone().numpy()
            o[f"uncert_{tag}_global_video"] = pu.get_global_uncert(vt.clone()).numpy()
    nb = 7
    ctr = np.stack([r.uniform(100, 1800, nb), r.uniform(50, 1000, nb)], 1)
    scl = r.uniform(0.6, 3.5, nb)
    shp = np.array([[1080, 1920], [720, 1280], [1080, 1920], [480, 640], [2160, 3840], [1080, 1920], [1000, 1000]], np.float64)
    o["bbinfo_center"], o["bbinfo_scale"], o["bbinfo_shape"] = ctr, scl, shp
    o["bbinfo_out"] = np.stack([hu["image_utils"].calculate_bbox_info(c, s_, sh) for c, s_, sh in zip(ctr, scl, shp)])
    o["bbinfo_focal"] = np.array([hu["image_utils"].calculate_focal_length(sh[0], sh[1]) for sh in shp])
    bbox = np.stack([ctr[:, 0], ctr[:, 1], scl * 200.0, scl * 200.0], 1).astype(np.float32)
    camc = np.stack([r.uniform(0.5, 1.2, nb), r.uniform(-.3, .3, nb), r.uniform(-.3, .3, nb)], 1).astype(np.float32)
    kp = r.uniform(-1.2, 1.2, (nb, 49, 2)).astype(np.float32)
    o["ccam_bbox"], o["ccam_cam"], o["ccam_kp"] = bbox, camc, kp
    o["ccam_orig_cam"] = hu["demo_utils"].convert_crop_cam_to_orig_img(camc, bbox, 1920, 1080)
    o["ccam_orig_kp"] = hu["demo_utils"].convert_crop_coords_to_orig_img(bbox, kp.copy(), 224)
    C = hu["constants"]
    o["joint_map"] = np.array([C.JOINT_MAP[n] for n in C.JOINT_NAMES], np.int32)       # smpl_head.py:17
    assert np.array_equal(o["joint_map"], synth.JOINT_MAP_49)
    o["img_norm_mean"], o["img_norm_std"] = np.array(C.IMG_NORM_MEAN), np.array(C.IMG_NORM_STD)
    # crop: the float32 point triples the reference's gen_trans_from_patch_cv hands to cv2.getAffineTransform
    # (vibe_image_utils.py:58-92; cv2 replaced by a recorder).  float64 boxes: independent of the numpy version's
    # scalar promotion (the reference pins numpy 1.18 where float32 * Python float is float64 anyway).
    import importlib
    vi = importlib.import_module("pocolib.utils.vibe_image_utils")
    seen = {}

    def _record(src, dst):
        seen["src"], seen["dst"] = np.array(src), np.array(dst)
        return np.zeros((2, 3))

    vi.cv2.getAffineTransform = _record
    rc = np.random.default_rng(4242)
    cb = np.stack([rc.uniform(-50, 2000, 12), rc.uniform(-50, 1100, 12), rc.uniform(8, 900, 12), rc.uniform(8, 900, 12)], 1)
    cb[0] = (112, 112, 224, 224)
    cb[1] = np.float32([100.5, 77.25, 33.3, 51.7]).astype(np.float64)
    csc = np.array([1.0, 1.0, 1.1, 1.2] * 3)
    srcs, dsts = [], []
    for (cx_, cy_, w_, h_), sc_ in zip(cb, csc):
        vi.gen_trans_from_patch_cv(cx_, cy_, w_, h_, 224, 224, float(sc_), 0, inv=False)
        assert seen["src"].dtype == np.float32 and seen["dst"].dtype == np.float32
        srcs.append(seen["src"]); dsts.append(seen["dst"])
    o["crop_boxes"], o["crop_scale"], o["crop_src"], o["crop_dst"] = cb, csc, np.stack(srcs), np.stack(dsts)
    from oracle import crop_np
    for bx, sc_, s_, d_ in zip(cb, csc, srcs, dsts):
        ps, pd = crop_np.patch_points(*bx, 224, sc_)
        assert np.array_equal(ps, s_) and np.array_equal(pd, d_)
    np.savez_compressed(GOLD / "ops.npz", **o)
    # pin the oracle's small functions
    assert np.abs(poco_ref.rot6d_to_rotmat(torch.from_numpy(x6)).numpy() - o["rot6d_out"]).max() < 1e-6
    assert np.abs(poco_ref.keypoint_attention(feat, heat).numpy() - o["ka_out"]).max() < 1e-5
    assert np.abs(poco_ref.kinematic_uncert(var, synth.SMPL_PARENTS) - v2).max() == 0
    assert np.abs(poco_ref.weak_persp_to_persp(cam).numpy() - o["cam_t"]).max() < 1e-6
    assert np.abs(poco_ref.project(pts, t, 5000.0, 0.0, 0.0).numpy() - o["cam_proj"]).max() < 2e-3
    assert np.abs(poco_ref.full_img_cam(cam, bt["scale"] * 200.0, bt["center"], bt["orig_shape"][:, 1],
                                        bt["orig_shape"][:, 0], bt["focal_length"]).numpy() - o["cam_full_t"]).max() < 1e-4
    print("ops fixtures written; oracle small-function pins OK")


def run_smooth():
    """tests/golden/smooth.npz: the reference's OneEuroFilter (one_euro_filter.py) driven exactly like
    smooth_pose.py:28-61 on a seeded rotation-matrix track; pins oracle/smooth_np.smooth_rotmats."""
    import importlib.util
    from oracle import smooth_np
    spec = importlib.util.spec_from_file_location(
        "ref_one_euro", os.path.join(ref_import.REFERENCE, "pocolib/utils/one_euro_filter.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    r = np.random.default_rng(2024)
    T = 17
    # a smooth random walk of rotations + jitter, as rotation matrices [T,24,3,3]
    aa = np.cumsum(0.05 * r.standard_normal((T, 24, 3)), 0) + 0.02 * r.standard_normal((T, 24, 3))
    th = np.linalg.norm(aa, axis=-1, keepdims=True) + 1e-12
    k = aa / th
    K = np.zeros((T, 24, 3, 3))
    K[..., 0, 1], K[..., 0, 2], K[..., 1, 0] = -k[..., 2], k[..., 1], k[..., 2]
    K[..., 1, 2], K[..., 2, 0], K[..., 2, 1] = -k[..., 0], -k[..., 1], k[..., 0]
    pose = (np.eye(3) + np.sin(th)[..., None] * K + (1 - np.cos(th))[..., None] * (K @ K)).astype(np.float32)
    out = {}
    for tag, (mc, be) in {"default": (0.004, 0.7), "demo": (0.004, 1.5)}.items():   # smooth_pose.py:25, demo.py:289-292
        f = m.OneEuroFilter(np.zeros_like(pose[0]), pose[0], min_cutoff=mc, beta=be)
        hat = np.zeros_like(pose)
        hat[0] = pose[0]
        for idx in range(1, T):
            hat[idx] = f(np.ones_like(pose[idx]) * idx, pose[idx])
        out[f"hat_{tag}"] = hat
        assert np.abs(smooth_np.smooth_rotmats(pose, mc, be) - hat).max() < 1e-6
    out["pose"] = pose
    np.savez_compressed(GOLD / "smooth.npz", **out)
    print("smooth fixture written; oracle pinned")


def main():
    assert ref_import.available(), "needs /root/reference"
    GOLD.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(8)
    mp = synth.synth_state_dict([("head.init_pose", (1, 144)), ("head.init_shape", (1, 10)), ("head.init_cam", (1, 3))], 0)
    ref_import.setup({"pose": mp["head.init_pose"][0], "shape": mp["head.init_shape"][0], "cam": mp["head.init_cam"][0]})
    only = sys.argv[1:]
    if not only or "ops" in only:
        run_ops()
    if not only or "smooth" in only:
        run_smooth()
    for v, kw in VARIANTS.items():
        if "calibrate" in only:
            calibrate(v, kw)
        if not only or v in only or "models" in only:
            run_variant(v, kw)
        if not only or v in only or "stress" in only or "calibrate" in only:
            run_variant(v, kw, "stress")


if __name__ == "__main__":
    main()
