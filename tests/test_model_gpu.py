"""GPU parity of the whole hot path (through the C ABI) against
  (a) the committed golden vectors produced by the reference's own modules (B=2), and
  (b) the CPU oracle on the same seeded inputs at other batch sizes.
Gate (BASELINE.json north_star): max-abs < 1e-3 on pose / shape / cam / confidence / vertices."""
import numpy as np
import pytest
import torch

from poco_amd import synth
from tests import util

pytestmark = pytest.mark.gpu
TOL = 1e-3
VARIANTS = ["hrnet_w32-pare", "hrnet_w48_cls-cliff", "resnet50-cliff"]


def _np(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("profile", ["default", "stress"])
@pytest.mark.parametrize("variant", VARIANTS)
def test_golden_b2(variant, profile, cuda):
    """Fixtures made by the reference's own modules.  profile "stress" = every BN gamma in [0.5,1.5] + structured
    crops (poco_amd/synth.py): the two crops of the fixture differ by >= 44x the gate on every compared output."""
    g = dict(np.load(util.GOLD / (f"model_{variant}.npz" if profile == "default" else f"model_{variant}_stress.npz")))
    m = util.make_engine(variant, max_batch=4, profile=profile)
    out = m(util.cuda_batch(synth.synth_batch(2, 1234, profile=profile), cuda))
    torch.cuda.synchronize()
    errs = {k: float(np.abs(_np(out[k]) - g[k]).max()) for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose")}
    p6 = out["pred_pose6d"] if "pred_pose6d" in out else out["pred_pose_6d"]
    errs["pred_pose6d"] = float(np.abs(_np(p6).reshape(2, -1) - g["pred_pose6d"]).max())
    errs["uncert_feat"] = float(np.abs(_np(out["uncert_feat"])[:, g["uncert_feat_idx"]] - g["uncert_feat_samples"]).max())
    errs["vertices"] = float(np.abs(_np(out["smpl_vertices"])[:, g["oracle_vert_idx"]] - g["oracle_vert_samples"]).max())
    errs["joints3d"] = float(np.abs(_np(out["smpl_joints3d"]) - g["oracle_smpl_joints3d"]).max())
    errs["cam_t"] = float(np.abs(_np(out["pred_cam_t"]) - g["oracle_pred_cam_t"]).max())
    j2 = np.abs(_np(out["smpl_joints2d"]) - g["oracle_smpl_joints2d"]).max()
    # full-image pixels (values ~1e3) for cliff: relative gate; crop-normalised for pare: absolute
    errs["joints2d"] = float(j2 / max(1.0, np.abs(g["oracle_smpl_joints2d"]).max()))
    if variant.endswith("cliff"):
        errs["body_feat2"] = float(np.abs(_np(out["body_feat2"])[:, :64] - g["body_feat2_samples"]).max())
        errs["fullimg_cam_t"] = float(np.abs(_np(out["pred_fullimg_cam_t"]) - g["oracle_pred_fullimg_cam_t"]).max()
                                      / max(1.0, np.abs(g["oracle_pred_fullimg_cam_t"]).max()))
    else:
        seg = _np(out["pred_segm_mask"]).reshape(2, -1)[:, g["segm_idx"]]
        errs["segm"] = float(np.abs(seg - g["segm_samples"]).max())
    print(variant, profile, errs)
    if profile == "stress":       # features are O(1..10) here: gate them relative to their size
        for k, idx, samp in (("uncert_feat", "uncert_feat_idx", "uncert_feat_samples"),):
            errs[k] /= max(1.0, float(np.abs(g[samp]).max()))
        if "body_feat2" in errs:
            errs["body_feat2"] /= max(1.0, float(np.abs(g["body_feat2_samples"]).max()))
        if "segm" in errs:
            errs["segm"] /= max(1.0, float(np.abs(g["segm_samples"]).max()))
    assert max(errs.values()) < TOL, errs
    assert out["log_phi"] is None and out["gt_pose_cond_idx"] == []


@pytest.mark.parametrize("profile", ["default", "stress"])
def test_golden_backbone_map_w32(profile, cuda):
    """HRNet-W32's 480-channel output map itself (hrnet.py:515-519: trunk + bilinear x2 chains + concat) against the samples, sum and
    mean magnitude the REFERENCE's backbone produced (tests/golden/model_hrnet_w32-pare*.npz: feat_*).  VERDICT r1 row a2: the map was
    only checked through what the head makes of it.  The engine copies it out on request (poco_outputs_t.backbone_feat)."""
    g = dict(np.load(util.GOLD / ("model_hrnet_w32-pare.npz" if profile == "default" else "model_hrnet_w32-pare_stress.npz")))
    m = util.make_engine("hrnet_w32-pare", max_batch=4, profile=profile)
    out = m._alloc_outputs(2, want_segm=False, want_backbone_feat=True)
    m(util.cuda_batch(synth.synth_batch(2, 1234, profile=profile), cuda), out=out)
    torch.cuda.synchronize()
    f = _np(out["backbone_feat"]).reshape(2, -1)
    assert f.shape[1] == 480 * 56 * 56
    scale = max(1.0, float(np.abs(g["feat_samples"]).max()))
    assert np.abs(f[:, g["feat_idx"]] - g["feat_samples"]).max() < TOL * scale
    assert np.abs(np.abs(f).mean(1) - g["feat_abs_mean"]).max() < TOL * scale
    assert np.abs(f.astype(np.float64).sum(1) - g["feat_sum"]).max() < TOL * scale * 1e3     # 1.5e6 terms per crop
    assert float(np.abs(g["feat_samples"][0] - g["feat_samples"][1]).max()) > 10 * TOL          # the two crops do differ


@pytest.mark.parametrize("variant,B", [("hrnet_w32-pare", 5), ("hrnet_w48_cls-cliff", 7), ("resnet50-cliff", 9),
                                       ("resnet50-cliff", 1)])
def test_oracle_other_batches(variant, B, cuda):
    """Ragged batch sizes (partial tiles in every kernel) against the CPU oracle; also checks that a
    smaller batch on a bigger workspace and repeated calls give identical results."""
    torch.set_num_threads(8)
    bnp = synth.synth_batch(B, 4321 + B)
    ref = util.oracle_forward(variant, bnp)
    m = util.make_engine(variant, max_batch=12)
    batch = util.cuda_batch(bnp, cuda)
    out = m(batch)
    out2 = m(batch)
    torch.cuda.synchronize()
    for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices", "smpl_joints3d", "pred_cam_t"):
        err = float(np.abs(_np(out[k]) - ref[k].numpy()).max())
        assert err < TOL, (k, err)
        assert torch.equal(out[k], out2[k]), k          # deterministic
    j2 = np.abs(_np(out["smpl_joints2d"]) - ref["smpl_joints2d"].numpy()).max() / max(1.0, ref["smpl_joints2d"].abs().max().item())
    assert j2 < TOL


@pytest.mark.parametrize("variant", ["hrnet_w32-pare", "hrnet_w48_cls-cliff"])
def test_lanes_equivalent(variant, cuda):
    """Multi-stream (forked lanes) and single-stream execution must give bit-identical outputs."""
    m = util.make_engine(variant, max_batch=6)
    batch = util.cuda_batch(synth.synth_batch(6, 99), cuda)
    m.set_num_lanes(1)
    a = {k: v.clone() for k, v in m(batch).items() if isinstance(v, torch.Tensor)}
    for lanes in (4, 2):
        m.set_num_lanes(lanes)
        for _ in range(2):
            b = m(batch)
            torch.cuda.synchronize()
            for k, v in a.items():
                assert torch.equal(v, b[k]), (lanes, k)


@pytest.mark.parametrize("variant", ["hrnet_w48_cls-cliff", "resnet50-cliff"])
def test_kconcat_shortcut_matches_separate_convs(variant, cuda, monkeypatch):
    """layer1.0: bn3(conv3(t)) + bn_d(conv_d(x)) as ONE 1x1 conv over [t ; x] (engine.hip bottleneck(), cat mode) against
    the two-launch form of the reference (hrnet.py:79-99 / resnet.py:101-121); only the summation order differs."""
    batch = util.cuda_batch(synth.synth_batch(5, 31), cuda)
    merged = util.make_engine(variant, max_batch=5)
    separate = util.make_engine(variant, max_batch=5, options={"kcat": 0})
    names = lambda m: [n for n, _, _ in m.ops()]
    assert any(n.endswith("layer1.0.conv3+downsample") for n in names(merged))
    assert any(n.endswith("layer1.0.downsample.0") for n in names(separate)) and len(names(separate)) == len(names(merged)) + 1
    a, b = merged(batch), separate(batch)
    for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices"):
        assert (a[k] - b[k]).abs().max().item() < 2e-5, k


def test_strided_kconcat_shortcut_matches_separate_convs(cuda, monkeypatch):
    """ResNet-50 layer2-4 .0: bn3(conv3(t)) + bn_d(conv_d(x, stride 2)) as one GEMM with two B-operand sources
    (gemm1x1.hip, launch_gemm1x1_dual) against the two-launch form of resnet.py:101-121."""
    variant = "resnet50-cliff"
    batch = util.cuda_batch(synth.synth_batch(5, 41), cuda)
    merged = util.make_engine(variant, max_batch=5)
    separate = util.make_engine(variant, max_batch=5, options={"dual": 0})
    names = lambda m: [n for n, _, _ in m.ops()]
    assert sum(n.endswith(".0.conv3+downsample") for n in names(merged)) == 4          # layer1.0 (K-concat buffer) + layer2-4.0
    assert len(names(separate)) == len(names(merged)) + 3
    a, b = merged(batch), separate(batch)
    for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices"):
        assert (a[k] - b[k]).abs().max().item() < 2e-5, k


@pytest.mark.parametrize("variant", ["hrnet_w32-pare", "resnet50-cliff"])
def test_chained_bottleneck_matches_separate_convs(variant, cuda, monkeypatch):
    """layer1: conv3 + residual + ReLU of block k and conv1 + ReLU of block k+1 as one kernel (csrc/bneck_chain.hip: the
    256-channel tensor stays in registers between the two GEMMs) against the two-launch form (hrnet.py:79-99 /
    resnet.py:101-121).  Ragged batch: 7 crops = 21952 pixels = 1372 sub-tiles, not a multiple of the 2048 waves."""
    batch = util.cuda_batch(synth.synth_batch(7, 77), cuda)
    chained = util.make_engine(variant, max_batch=7)
    separate = util.make_engine(variant, max_batch=7, options={"chain": 0})
    names = lambda m: [n for n, _, _ in m.ops()]
    nchain = sum("conv3+" in n and n.endswith(".conv1") for n in names(chained))
    assert nchain == (2 if variant.startswith("hrnet") else 1) and len(names(separate)) == len(names(chained)) + nchain
    a, b = chained(batch), separate(batch)
    for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices"):
        assert (a[k] - b[k]).abs().max().item() < 2e-5, k


@pytest.mark.parametrize("variant,B", [("resnet50-cliff", 1), ("resnet50-cliff", 23), ("hrnet_w48_cls-cliff", 64), ("resnet50-cliff", 130)])
def test_fused_regressor_matches_separate_launches(variant, B, cuda):
    """cliff head: init state + fc1 / fc2 / decoders x 3 iterations + state scatter + rot6d (cliff_head.py:96-118) as ONE persistent
    launch with grid barriers between the stages (csrc/mlp_chain.hip) against the chain of ~20 separate launches (option
    mlp_fuse=0).  The K-split and the order of the partial sums are those of linear_mfma_kernel wherever that one splits K over
    8 waves, so most outputs agree bitwise; the gate is 2e-5.  Ragged batches (23, 130 crops = 2 / 9 row tiles), one crop, and the
    result must not depend on the grid (mlp_blocks 1 / 64 / 256) nor change over replays (the kernel re-arms its own counters)."""
    batch = util.cuda_batch(synth.synth_batch(B, 31), cuda)
    fused = util.make_engine(variant, max_batch=B, profile="stress")
    separate = util.make_engine(variant, max_batch=B, profile="stress", options={"mlp_fuse": 0})
    names = lambda m: [n for n, _, _ in m.ops()]
    assert "head.regressor" in names(fused) and "head.regressor" not in names(separate)
    assert len(names(separate)) - len(names(fused)) == 18
    keys = ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices", "uncert_feat", "body_feat2")
    a = {k: v.clone() for k, v in fused(batch).items() if isinstance(v, torch.Tensor)}
    b = separate(batch)
    for k in keys:
        assert (a[k] - b[k]).abs().max().item() < 2e-5 * max(1.0, b[k].abs().max().item()), k
    for _ in range(3):                       # replays: same counters, same results
        c = fused(batch)
        for k in keys:
            assert torch.equal(a[k], c[k]), k
    for nb in (1, 64, 256):
        other = util.make_engine(variant, max_batch=B, profile="stress", options={"mlp_blocks": nb})
        c = other(batch)
        for k in keys:
            assert torch.equal(a[k], c[k]), (k, nb)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("B,blocks", [(5, 256), (64, 256), (64, 7)])
def test_timed_out_in_kernel_wait_is_reported_and_the_engine_recovers(B, blocks, cuda):
    """VERDICT r5 weak #5 / ADVICE r5: the fused regressor's grid barrier (csrc/mlp_chain.hip) is a bounded poll; a forward whose wait ran
    out used to return POCO_OK with invalid outputs and leave the engine refusing every later forward.  Build option
    debug_mlp_timeouts=2 makes the first two launches of the regressor leave one block out of their first barrier (debug_wait_spins
    bounds the poll to ~1 ms).  Expected: nothing hangs; poco_forward itself still returns OK (it only enqueues); after a stream
    synchronise poco_status reports the failure ONCE and clears it; a forward enqueued on top of an unreported failure is refused;
    after the report the engine runs normally and its outputs are bitwise those of an engine that never failed - also through a
    hipGraph replay."""
    from poco_amd._lib import PocoHipError
    variant = "resnet50-cliff"
    batch = util.cuda_batch(synth.synth_batch(B, 5), cuda)
    good = util.make_engine(variant, max_batch=B, profile="stress", options={"mlp_blocks": blocks})
    ref = {k: v.clone() for k, v in good(batch).items() if isinstance(v, torch.Tensor)}
    good.check_status(sync=True)                                  # the good path: no error, cheap
    m = util.make_engine(variant, max_batch=B, profile="stress", options={"mlp_blocks": blocks, "debug_wait_spins": 2000, "debug_mlp_timeouts": 2})
    keys = ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices", "record")
    for attempt in range(2):
        out = m(batch)                                            # enqueues fine: the failure happens on the device, later
        torch.cuda.synchronize()
        if blocks > 1:
            with pytest.raises(PocoHipError, match="INVALID"):
                m.check_status()
        m.check_status()                                          # reported once, cleared
    # a forward on top of an UNREPORTED failure is refused (for callers that never ask)
    m2 = util.make_engine(variant, max_batch=B, profile="stress", options={"mlp_blocks": blocks, "debug_wait_spins": 2000, "debug_mlp_timeouts": 1})
    m2(batch)
    torch.cuda.synchronize()
    with pytest.raises(PocoHipError, match="poco_status"):
        m2(batch)
    with pytest.raises(PocoHipError, match="INVALID"):
        m2.check_status()
    for eng in (m, m2):                                           # re-armed: normal forwards, eager and as a graph replay
        out = eng(batch)
        eng.check_status(sync=True)
        for k in keys:
            assert torch.equal(out[k], ref[k]), k
        o2 = eng._alloc_outputs(B, want_segm=False)
        for _ in range(2):
            eng.graph_forward(batch, o2)
        eng.check_status(sync=True)
        for k in keys:
            assert torch.equal(o2[k], ref[k]), k


@pytest.mark.parametrize("variant,B", [("resnet50-cliff", 3), ("hrnet_w32-pare", 5), ("resnet50-cliff", 64), ("hrnet_w48_cls-cliff", 17)])
def test_mfma_stem_matches_valu_stem(variant, B, cuda):
    """Stem conv (7x7 / 3x3, stride 2, Cin = 3; resnet.py:203-205, hrnet.py:467-469) as an implicit GEMM on the fp32 MFMA
    (csrc/stem_mfma.hip: one output row per wave, zero-padded input rows double-buffered in LDS) against the packed-FMA kernels
    (option stem_mfma=0).  Different summation order over the 147 / 27 taps: the gate is 2e-5 on the regressed outputs.  Batches
    that give every block 1, 2 and 7 chunks of rows, with blocks that end past the last image row (17 crops: 5 chunks x 6 blocks)."""
    batch = util.cuda_batch(synth.synth_batch(B, 13), cuda)
    mfma = util.make_engine(variant, max_batch=B, profile="stress")
    valu = util.make_engine(variant, max_batch=B, profile="stress", options={"stem_mfma": 0})
    a, b = mfma(batch), valu(batch)
    for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices"):
        assert (a[k] - b[k]).abs().max().item() < 2e-5 * max(1.0, b[k].abs().max().item()), k


GATED = ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices", "smpl_joints3d")   # north_star: abs 1e-3
FEATS = ("uncert_feat", "body_feat2", "pred_segm_mask")                                            # relative 1e-3


def bench_batch_deviation(variant, B, cuda, profile="stress", seed=2024, pick=None, engine=None):
    """Engine (tuned table applied for this batch size) vs the CPU oracle on `pick` crops of a B-crop batch.
    Returns ({key: max-abs deviation}, {key: inter-crop spread of the reference}, {key: max |ref|})."""
    torch.set_num_threads(16)
    bnp = synth.synth_batch(B, seed, profile=profile)
    # the stress profile checks EVERY crop of the batch against the oracle (4-16 s of oracle per case) - fuse sums,
    # avg-pool, the regressor's row tiles, LBS crop blocks and record packing are then covered at the bench sizes crop by crop, not
    # only through property checks; the default profile (crops differ by ~1e-3 there) keeps the 6-crop pick
    if pick is None:
        pick = np.arange(B) if profile == "stress" else np.r_[0:3, B - 3:B]
    pick = np.asarray(pick)
    ref = util.oracle_forward(variant, {k: v[pick] for k, v in bnp.items()}, profile=profile)
    m = engine or util.make_engine(variant, max_batch=B, profile=profile)
    out = m(util.cuda_batch(bnp, cuda))
    torch.cuda.synchronize()
    dev, spread, mag = {}, {}, {}
    for k in GATED + FEATS:
        if k not in ref or k not in out:
            continue
        r = ref[k].numpy()
        dev[k] = float(np.abs(_np(out[k])[pick] - r).max())
        spread[k] = float(np.abs(r[0] - r[1]).max())
        mag[k] = float(np.abs(r).max())
    return dev, spread, mag


@pytest.mark.parametrize("profile", ["stress", "default"])
@pytest.mark.parametrize("variant,B", [("hrnet_w48_cls-cliff", 64), ("hrnet_w32-pare", 32), ("resnet50-cliff", 64),
                                       ("hrnet_w48_cls-cliff", 128)])
def test_bench_batch_with_tuned_table(variant, B, profile, cuda):
    """The batch sizes bench.py runs use the measured tile table (Winograd F(2x2)/F(4x4), LDS-DMA, persistent variants,
    poco_amd/tuned/gfx950.json).  Same 1e-3 abs gate on the north_star outputs; backbone/head features
    (uncert_feat, body_feat2, pred_segm_mask) within 1e-3 of their magnitude.  With the stress profile (undamped
    residual branches) the test also asserts it is not vacuous: the reference's crop-to-crop variation
    of every compared output is >= 10x the tolerance applied to it."""
    from poco_amd import tune
    assert any(k.startswith(f"{B}x") for k in tune.load_table()), "tuned table missing for the bench batch size"
    dev, spread, mag = bench_batch_deviation(variant, B, cuda, profile)
    print(variant, B, profile, "max-abs deviation with tuned kernels:", {k: "%.2e" % v for k, v in dev.items()},
          "| inter-crop spread:", {k: "%.2e" % v for k, v in spread.items()})
    for k, e in dev.items():
        tol = TOL if k in GATED else TOL * max(1.0, mag[k])
        assert e < tol, (k, e, tol)
        if profile == "stress":
            assert spread[k] >= 10 * tol, ("vacuous comparison", k, spread[k], tol)


@pytest.mark.parametrize("variant,B", [("hrnet_w48_cls-cliff", 9), ("hrnet_w32-pare", 5)])
def test_wino4g_chained_convs_match_unchained(variant, B, cuda, monkeypatch):
    """ALG 11 (Winograd F(4x4) as position GEMMs) on the 7x7 branch: consecutive convs of the chain run with the fused tail
    (wg_mid_kernel: output transform of conv k -> V of conv k+1 through LDS, conv1's output of a BasicBlock never stored) -
    BITWISE the same model outputs as the unchained three-launch form (option wg_fuse=0), in eager 4-lane, 1-lane and
    graph-replay mode, and within the gate of the oracle."""
    batch_np = synth.synth_batch(B, 21)
    batch = util.cuda_batch(batch_np, cuda)
    keys = ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices", "uncert_feat")

    def engine(fuse):
        m = util.make_engine(variant, max_batch=B, profile="stress", options={"wg_fuse": int(fuse)})
        n = 0
        for i, _ in enumerate(m.ops()):
            d = m.conv_desc(i)
            if d is not None and d[4] == 3 and d[5] == 1 and d[0] <= 8 and d[1] <= 8 and d[0] * d[1] > 1:
                m.set_conv_cfg(i, B, (2, 4, 2, 2, 3, 1, 11))
                n += 1
        assert n >= 24
        return m

    plain = engine(False)
    ref = {k: v.clone() for k, v in plain(batch).items() if k in keys}
    fused = engine(True)
    for lanes in (4, 1):
        fused.set_num_lanes(lanes)
        out = fused(batch)
        for k in keys:
            assert torch.equal(out[k], ref[k]), (k, lanes)
    fused.set_num_lanes(4)
    o = fused._alloc_outputs(B, False)
    fused.graph_forward(batch, o)
    fused.graph_forward(batch, o)
    torch.cuda.synchronize()
    for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices"):
        assert torch.equal(o[k], ref[k]), (k, "graph")
    orc = util.oracle_forward(variant, batch_np, profile="stress")
    for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices"):
        assert np.abs(_np(ref[k]) - orc[k].numpy()).max() < TOL, k


@pytest.mark.parametrize("variant,B", [("hrnet_w48_cls-cliff", 5), ("hrnet_w32-pare", 3)])
def test_kmerged_fuse_convs_match_separate_form(variant, B, cuda, monkeypatch):
    """K-merge (engine.hip hr_module): relu(x_T + sum_j bn_j(conv_j(t_j))) of the lowest-resolution branch of every HR module
    (hrnet.py:208-236, 248-266) runs as ONE stride-2 conv over the channel concat with residual + ReLU in the epilogue.  Against
    the separate form (option kmerge=0: T convs + fuse_sum): 8 fuse_sum launches and 12 (W48) conv launches fewer, the same
    parameters consumed, outputs equal up to the summation order (fp32 rounding, far inside the gate) on the STRESS weights,
    and both inside the gate of the oracle."""
    batch_np = synth.synth_batch(B, 33)
    batch = util.cuda_batch(batch_np, cuda)
    keys = ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices")

    def engine(merge):
        return util.make_engine(variant, max_batch=B, profile="stress", options={"kmerge": int(merge)})

    sep = engine(False)
    mrg = engine(True)
    names_sep = [o[0] for o in sep.ops()]
    names_mrg = [o[0] for o in mrg.ops()]
    n_mod = 8                                                     # 1 + 4 + 3 modules
    assert sum(n.endswith("+down") for n in names_mrg) == n_mod and not any(n.endswith("+down") for n in names_sep)
    assert len(names_sep) - len(names_mrg) >= n_mod               # at least the fuse_sum of branch T per module
    ref = {k: v.clone() for k, v in sep(batch).items() if k in keys}
    out = {k: v.clone() for k, v in mrg(batch).items() if k in keys}
    orc = util.oracle_forward(variant, batch_np, profile="stress")
    for k in keys:
        d = float((out[k] - ref[k]).abs().max())
        assert d < 0.1 * TOL, (k, d)
        assert np.abs(_np(out[k]) - orc[k].numpy()).max() < TOL, k
        assert np.abs(_np(ref[k]) - orc[k].numpy()).max() < TOL, k
    # graph replay and single lane: bitwise the eager 4-lane result
    o = mrg._alloc_outputs(B, False)
    mrg.graph_forward(batch, o)
    mrg.graph_forward(batch, o)
    torch.cuda.synchronize()
    for k in keys:
        assert torch.equal(o[k], out[k]), (k, "graph")
    mrg.set_num_lanes(1)
    out1 = mrg(batch)
    for k in keys:
        assert torch.equal(out1[k], out[k]), (k, "1 lane")


@pytest.mark.parametrize("variant,B", [("hrnet_w48_cls-cliff", 6), ("hrnet_w32-pare", 4)])
def test_one_join_per_module_schedule_is_bitwise_the_three_join_one(variant, B, cuda, monkeypatch):
    """Scheduling only (engine.hip hr_module / hrnet_trunk, `xdep`): the fuse convs on the lane of the branch they read, one
    join per HR module instead of three, stage boundaries and the cls-head incre modules inside the open region with the
    transition conv waiting for the K-merged conv's lane through an event.  Same kernels, same configurations, same operands:
    every output must be BITWISE what the three-join schedule (option xdep=0) produces - eager on 4 lanes, on 1 lane, and as a
    replayed hipGraph (repeated: a missing dependency shows up as a rare differing tile)."""
    batch = util.cuda_batch(synth.synth_batch(B, 57), cuda)
    keys = ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices", "uncert_feat")
    old = util.make_engine(variant, max_batch=B, profile="stress", options={"xdep": 0})
    ref = {k: v.clone() for k, v in old(batch).items() if k in keys}
    new = util.make_engine(variant, max_batch=B, profile="stress")
    assert [o[0] for o in new.ops()] != [] and len(new.ops()) == len(old.ops())
    for lanes in (4, 1, 4):
        new.set_num_lanes(lanes)
        for _ in range(5):
            out = new(batch)
            for k in keys:
                assert torch.equal(out[k], ref[k]), (k, lanes)
    o = new._alloc_outputs(B, False)
    for _ in range(10):
        new.graph_forward(batch, o)
        torch.cuda.synchronize()
        for k in keys[:-1]:
            assert torch.equal(o[k], ref[k]), (k, "graph")


def test_split_f16_experiment_passes_the_gate(cuda, monkeypatch):
    """VERDICT r2 next #9 (EXPERIMENT, bench.py --split-f16, never the default): ResNet-50-CLIFF with every plain 1x1 conv on the
    split-fp16 GEMM (fp16 hi + lo, 3 MFMAs per product) must pass the STRESS fixtures at the same 1e-3 gate - golden B = 2 made by
    the reference's modules, and the oracle at the bench batch with the tuned table."""
    from tests import util as _u
    if not _u.has_experiments():
        pytest.skip("experiment build only (python -m poco_amd.build --experiments; POCO_HIP_LIB=poco_amd/lib/exp/libpoco_hip_experiments.so)")
    variant = "resnet50-cliff"
    m = util.make_engine(variant, max_batch=2, profile="stress", options={"split_f16": 1})
    n12 = sum(1 for i, _ in enumerate(m.ops()) if m.conv_desc(i) is not None and m.conv_desc(i)[4] == 1 and m.conv_desc(i)[2] % 32 == 0
              and m.conv_desc(i)[0] * m.conv_desc(i)[1] >= 16)
    assert n12 >= 20
    g = np.load(util.GOLD / f"model_{variant}_stress.npz")
    out = m(util.cuda_batch(synth.synth_batch(2, 1234, profile="stress"), cuda))
    worst = {}
    for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose"):
        worst[k] = float(np.abs(_np(out[k]).reshape(g[k].shape) - g[k]).max())
    dev, spread, mag = bench_batch_deviation(variant, 64, cuda, "stress")
    print("split-f16 experiment: golden B=2", {k: "%.1e" % v for k, v in worst.items()}, "| oracle B=64", {k: "%.1e" % v for k, v in dev.items()})
    assert max(worst.values()) < TOL, worst
    for k, e in dev.items():
        assert e < (TOL if k in GATED else TOL * max(1.0, mag[k])), (k, e)


@pytest.mark.parametrize("variant,B", [("hrnet_w48_cls-cliff", 6), ("hrnet_w32-pare", 5), ("resnet50-cliff", 9)])
def test_record_output_is_the_packed_host_record(variant, B, cuda):
    """poco_outputs_t.record: the 254-float per-crop record - the payload of the multi-GPU all-gather and of the
    streaming D2H copy - is written by ONE kernel inside the forward.  Against dist.pack_records (the torch restatement of
    poco_utils.py:21-25,50-60 + tester.py:245 used since round 1): the 253 copied floats BITWISE, the post-processed confidence
    bitwise for the CLIFF rule (root value) and to one ulp for PARE's mean over 24 joints (summation order); eager, graph replay
    and with the post-processing options of poco_create_ex; uncertainties scaled so that both sides of the threshold occur."""
    from poco_amd import dist as pdist
    from poco_amd import postproc
    batch = util.cuda_batch(synth.synth_batch(B, 12), cuda)
    m = util.make_engine(variant, max_batch=B, profile="stress")
    out = m(batch)
    rec = out["record"]
    ref = pdist.pack_records(out, head=variant)
    assert rec.shape == (B, 254) and torch.equal(rec[:, :253], ref[:, :253])
    assert (rec[:, 253] - ref[:, 253]).abs().max().item() <= (0.0 if "cliff" in variant else 1e-6)
    # the numpy post-processing the testers use (pinned to the reference's POCOUtils in tests/test_host_cpu.py)
    _, g = postproc.folder_uncert(out["var_pose"], variant, True)
    assert np.abs(_np(rec[:, 253]) - g).max() <= 1e-6
    o = m._alloc_outputs(B, False)
    m.graph_forward(batch, o)
    m.graph_forward(batch, o)
    torch.cuda.synchronize()
    assert torch.equal(o["record"], rec)
    # options: no kinematic accumulation, another threshold (both branches of the threshold rule exercised)
    var = out["var_pose"]
    root = float(var[:, 0].median())
    for kin in (0, 1):
        m2 = util.make_engine(variant, max_batch=B, profile="stress", options={"record_kinematic": kin, "record_thr": root if "pare" in variant else root / 2})
        o2 = m2(batch)
        want = pdist.global_confidence(o2["var_pose"], variant, kinematic=bool(kin), thr=root if "pare" in variant else root / 2)
        assert (o2["record"][:, 253] - want).abs().max().item() <= 1e-6
        hot = o2["var_pose"][:, 0] > root
        assert bool(hot.any()) and bool((~hot).any())
        assert torch.equal(o2["record"][:, :253], rec[:, :253])
    # a caller that does not ask for the record gets the same outputs (the op is skipped)
    o3 = {k: v for k, v in m._alloc_outputs(B, False).items() if k != "record"}
    m(batch, out=o3)
    assert torch.equal(o3["pred_pose"], out["pred_pose"])


def test_graph_cache_is_bounded(cuda):
    """graph_forward keeps at most max_graphs captured graphs (LRU); release_graphs() drops them."""
    m = util.make_engine("resnet50-cliff", max_batch=2)
    m.max_graphs = 2
    batch = util.cuda_batch(synth.synth_batch(2, 3), cuda)
    outs = [m._alloc_outputs(2, False) for _ in range(4)]
    for o in outs:
        m.graph_forward(batch, o)
        assert len(m._graphs) <= 2
    first = m(batch)
    m.graph_forward(batch, outs[0])                      # evicted earlier: re-captured, still correct
    torch.cuda.synchronize()
    assert torch.equal(outs[0]["pred_pose"], first["pred_pose"]) and len(m._graphs) == 2
    m.release_graphs()
    assert not hasattr(m, "_graphs") or not m._graphs


def test_graph_cache_tells_views_of_one_tensor_apart(cuda):
    """the cache key was (name, data_ptr) only, so x[:3] of tensors already captured at 6 crops replayed the
    6-crop graph and overwrote rows 3-5 of the caller's outputs.  The key now carries the shapes: the view gets its own capture."""
    m = util.make_engine("resnet50-cliff", max_batch=6)
    batch = util.cuda_batch(synth.synth_batch(6, 11), cuda)
    out = m._alloc_outputs(6, False)
    m.graph_forward(batch, out)
    torch.cuda.synchronize()
    full = {k: v.clone() for k, v in out.items()}
    # refill rows 3-5 with other crops, then ask for the first three only through views of the SAME storage
    other = util.cuda_batch(synth.synth_batch(6, 12), cuda)
    for k in batch:
        batch[k][3:] = other[k][3:]
    sentinel = {k: v[3:].clone().fill_(-7.0) for k, v in out.items()}
    for k, v in out.items():
        v[3:] = sentinel[k]
    vb = {k: v[:3] for k, v in batch.items()}
    vo = {k: v[:3] for k, v in out.items()}
    assert all(vb[k].data_ptr() == batch[k].data_ptr() for k in batch)
    m.graph_forward(vb, vo)
    torch.cuda.synchronize()
    assert len(m._graphs) == 2
    for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices", "record"):
        assert torch.equal(out[k][:3], full[k][:3]), k                     # crops are independent: rows 0-2 as before
        assert torch.equal(out[k][3:], sentinel[k]), k                     # rows 3-5 of the caller's storage untouched
    m.graph_forward(batch, out)                                            # the 6-crop graph is still there and sees the new rows
    torch.cuda.synchronize()
    assert len(m._graphs) == 2 and not torch.equal(out["pred_pose"][3:], full["pred_pose"][3:])
    assert torch.equal(out["pred_pose"][:3], full["pred_pose"][:3])


def test_graph_replay_matches_eager(cuda):
    """hipGraph replay of the forward (with the forked lanes captured) is bit-identical to eager launches,
    also after the inputs were refilled in place."""
    m = util.make_engine("hrnet_w48_cls-cliff", max_batch=4)
    b1 = util.cuda_batch(synth.synth_batch(4, 5), cuda)
    b2 = util.cuda_batch(synth.synth_batch(4, 6), cuda)
    ref1 = {k: v.clone() for k, v in m(b1).items() if isinstance(v, torch.Tensor)}
    ref2 = {k: v.clone() for k, v in m(b2).items() if isinstance(v, torch.Tensor)}
    out = m._alloc_outputs(4, False)
    m.graph_forward(b1, out)
    torch.cuda.synchronize()
    for k in ("pred_pose", "pred_shape", "smpl_vertices", "var_pose"):
        assert torch.equal(out[k], ref1[k]), k
    for k in b1:
        b1[k].copy_(b2[k])
    m.graph_forward(b1, out)
    torch.cuda.synchronize()
    for k in ("pred_pose", "pred_shape", "smpl_vertices", "var_pose"):
        assert torch.equal(out[k], ref2[k]), k


def test_smpl_lbs_op(cuda):
    """poco_smpl_lbs vs the float64 numpy restatement; identity pose + zero betas -> template."""
    from oracle import poco_ref, smpl_np
    m = util.make_engine("resnet50-cliff", max_batch=16)
    smpl = synth.synth_smpl(7)
    r = np.random.default_rng(5)
    B = 11
    betas = r.standard_normal((B, 10)).astype(np.float32)
    R = poco_ref.rot6d_to_rotmat(torch.from_numpy(r.standard_normal((B * 24, 6)).astype(np.float32))).reshape(B, 24, 3, 3)
    v64, j64 = smpl_np.smpl_lbs_np(smpl, betas, R.numpy())
    v, j = m.smpl_lbs(torch.from_numpy(betas).to(cuda), R.to(cuda))
    assert np.abs(_np(v) - v64).max() < 2e-5 and np.abs(_np(j) - j64).max() < 2e-5
    eye = torch.eye(3).repeat(2, 24, 1, 1).to(cuda)
    v0, _ = m.smpl_lbs(torch.zeros(2, 10, device=cuda), eye)
    assert np.abs(_np(v0) - smpl["v_template"][None]).max() < 1e-6


@pytest.mark.parametrize("variant,L", [("hrnet_w32-pare", 3), ("resnet50-cliff", 1)])
def test_realnvp_op(variant, L, cuda):
    """RealNVP log_prob / forward_p vs the oracle (itself pinned to the reference's RealNVP in
    tests/golden/ops.npz), plus the bijection property backward(forward(z)) == z via log_prob shift."""
    from oracle import poco_ref
    N = 48 + 5
    m = util.make_engine(variant, max_batch=2, options={"flow_ctx_rows": N})      # one context per row: scratch planned for N rows
    sd = poco_ref.to_torch(util.synth_weights(variant))
    r = np.random.default_rng(8)
    x = torch.from_numpy(np.abs(r.standard_normal((N, 9))).astype(np.float32))
    c = torch.from_numpy(r.standard_normal((N, 512)).astype(np.float32))
    ref_lp = poco_ref.realnvp_log_prob(sd, x, c).numpy()
    ref_fw = poco_ref.realnvp_forward(sd, x, c).numpy()
    lp = _np(m.realnvp_log_prob(x.to(cuda), c.to(cuda)))
    fw = _np(m.realnvp_forward(x.to(cuda), c.to(cuda)))
    assert np.abs(lp - ref_lp).max() < 1e-3 * max(1.0, np.abs(ref_lp).max())
    assert np.abs(fw - ref_fw).max() < 1e-3 * max(1.0, np.abs(ref_fw).max())
    back, _ = poco_ref.realnvp_backward(sd, torch.from_numpy(fw), c)
    assert np.abs(back.numpy() - x.numpy()).max() < 1e-3


@pytest.mark.parametrize("variant,L", [("hrnet_w32-pare", 3), ("resnet50-cliff", 1)])
@pytest.mark.parametrize("N", [1, 24, 1536, 3072, 24 * 128 * 8 + 5])
def test_realnvp_op_at_reference_sizes(variant, L, N, cuda):
    """the flow at the sizes the reference runs it - N = B*24 rows of 9 residuals with the crop's 512-d
    context repeated per joint (nf_head.py:93-110: bar_pose.reshape(-1, 9), repeat_interleave(context, 24)): B = 64 -> 1536,
    B = 128 -> 3072, 2*L = 2 and 6 coupling layers, plus 1 row, one crop's 24 rows and a ragged N far above any grid size.
    Against the oracle (pinned to the reference's RealNVP), rows bitwise independent of their neighbours."""
    from oracle import poco_ref
    from poco_amd._lib import PocoHipError
    m = util.make_engine(variant, max_batch=2, options={"flow_ctx_rows": N})      # (the per-row-context form needs N context rows)
    sd = poco_ref.to_torch(util.synth_weights(variant))
    r = np.random.default_rng(80 + N % 97)
    crops = (N + 23) // 24
    c = torch.from_numpy(np.repeat(r.standard_normal((crops, 512)).astype(np.float32), 24, axis=0)[:N].copy())
    x = torch.from_numpy(np.abs(r.standard_normal((N, 9))).astype(np.float32) * 1.5)
    ref_lp = poco_ref.realnvp_log_prob(sd, x, c).numpy()
    ref_fw = poco_ref.realnvp_forward(sd, x, c).numpy()
    xd, cd = x.to(cuda), c.to(cuda)
    lp = _np(m.realnvp_log_prob(xd, cd))
    fw = _np(m.realnvp_forward(xd, cd))
    assert lp.shape == (N,) and fw.shape == (N, 9)
    assert np.abs(lp - ref_lp).max() < 1e-3 * max(1.0, np.abs(ref_lp).max()), np.abs(lp - ref_lp).max()
    assert np.abs(fw - ref_fw).max() < 1e-3 * max(1.0, np.abs(ref_fw).max()), np.abs(fw - ref_fw).max()
    assert np.abs(ref_lp).max() > 1.0 and (N == 1 or np.std(ref_lp) > 1e-2)            # the comparison is not vacuous
    # the reference's call shape (one context per crop, used by its 24 rows) without the repeat: same rows, bitwise
    cc = cd[::24].contiguous()
    assert np.array_equal(_np(m.realnvp_log_prob(xd, cc, rep=24)), lp) or N > 256      # (the context GEMM kernel depends on the row count)
    assert np.abs(_np(m.realnvp_log_prob(xd, cc, rep=24)) - lp).max() < 1e-4 * max(1.0, np.abs(lp).max())
    assert np.abs(_np(m.realnvp_forward(xd, cc, rep=24)) - fw).max() < 1e-4 * max(1.0, np.abs(fw).max())
    if N >= 48:      # a row's result does not depend on which rows share its block (same GEMM kernel for <= 256 context rows)
        sub = slice(17, 41)
        a = _np(m.realnvp_log_prob(xd[sub].contiguous(), cd[sub].contiguous()))
        b = _np(m.realnvp_forward(xd[sub].contiguous(), cd[sub].contiguous()))
        if N <= 256:
            assert np.array_equal(a, lp[sub]) and np.array_equal(b, fw[sub])
        assert np.abs(a - lp[sub]).max() < 1e-4 * max(1.0, np.abs(lp).max()) and np.abs(b - fw[sub]).max() < 1e-4 * max(1.0, np.abs(fw).max())
    # SURVEY 8(b) "no allocation inside forward": the context-GEMM scratch is planned at finalize; a call with more context rows
    # than planned fails loudly instead of allocating
    small = util.make_engine(variant, max_batch=2)                                 # default plan: max_batch = 2 context rows
    if N > 2 * 24:
        with pytest.raises(PocoHipError, match="flow_ctx_rows"):
            small.realnvp_log_prob(xd, cc, rep=24)
    else:
        assert np.array_equal(_np(small.realnvp_log_prob(xd, cc, rep=24)), _np(m.realnvp_log_prob(xd, cc, rep=24)))


@pytest.mark.parametrize("variant,B", [("hrnet_w48_cls-cliff", 64), ("hrnet_w32-pare", 32), ("resnet50-cliff", 128)])
def test_full_size_properties(variant, B, cuda):
    """Size-independent properties at BASELINE.json's batch sizes (where the CPU oracle would take minutes):
    crops are independent, so (1) permuting the batch permutes the rows BITWISE (tiles of different crops share
    MFMA blocks but never mix), (2) duplicated crops give identical rows, (3) the returned mesh is exactly the
    stand-alone LBS operator applied to the returned parameters, (4) confidences are sigmoids, rotations
    orthonormal (rot6d_to_rotmat, geometry.py:247-261)."""
    m = util.make_engine(variant, max_batch=B)
    bnp = synth.synth_batch(B, 99)
    for k in bnp:                                  # duplicates: crop 1 := crop 0, last := first
        bnp[k][1] = bnp[k][0]
        bnp[k][B - 1] = bnp[k][0]
    batch = util.cuda_batch(bnp, cuda)
    out = {k: v.clone() for k, v in m(batch).items() if torch.is_tensor(v)}
    perm = torch.from_numpy(np.random.default_rng(3).permutation(B)).to(cuda)
    outp = m({k: v[perm].contiguous() for k, v in batch.items()})
    for k in ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices", "smpl_joints2d", "uncert_feat"):
        assert torch.equal(outp[k], out[k][perm]), k
        assert torch.equal(out[k][0], out[k][1]) and torch.equal(out[k][0], out[k][B - 1]), k
    v, j = m.smpl_lbs(out["pred_shape"], out["pred_pose"].contiguous())
    assert torch.equal(v, out["smpl_vertices"]) and torch.equal(j, out["smpl_joints3d"])
    R = out["pred_pose"].reshape(-1, 3, 3)
    eye = torch.eye(3, device=cuda).expand_as(R)
    assert float((R @ R.transpose(1, 2) - eye).abs().max()) < 1e-5 and float((torch.linalg.det(R) - 1).abs().max()) < 1e-5
    assert float(out["var_pose"].min()) > 0.0 and float(out["var_pose"].max()) < 1.0
    assert all(bool(torch.isfinite(t).all()) for t in out.values())


def test_repeatability_bitwise(cuda):
    """Race screen (short form of tools/stress.py): repeated forwards on the same inputs are bitwise identical in
    eager 4-lane, eager 1-lane and graph-replay mode (LDS-DMA / barrier / lane-join races would show up as rare
    differing tiles)."""
    m = util.make_engine("hrnet_w48_cls-cliff", max_batch=16)
    batch = util.cuda_batch(synth.synth_batch(16, 5), cuda)
    keys = ("pred_pose", "pred_shape", "pred_cam", "var_pose", "smpl_vertices", "uncert_feat")
    ref = {k: v.clone() for k, v in m(batch).items() if k in keys}
    out = m._alloc_outputs(16, False)
    for mode in ("eager4", "eager1", "graph"):
        m.set_num_lanes(1 if mode == "eager1" else 4)
        for _ in range(15):
            o = m.graph_forward(batch, out) if mode == "graph" else m(batch)
            for k in keys:
                assert torch.equal(o[k], ref[k]), (mode, k)
