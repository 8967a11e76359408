"""Parity of the MFMA conv+BN+ReLU(+residual) kernel against torch CPU conv2d (fp64 accumulate).

Call sites replaced: pocolib/models/backbone/hrnet.py:42-58,79-99 (conv->bn->relu, += residual).
Tolerance: 2e-5 * max|ref| (fp32 MFMA is an exact fmaf chain; only the summation order differs).
"""
import zlib

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _ref(x_nhwc, w, scale, shift, stride, res, relu):
    x = torch.from_numpy(x_nhwc).permute(0, 3, 1, 2).double()
    y = F.conv2d(x, torch.from_numpy(w).double(), stride=stride, padding=(w.shape[2] - 1) // 2)
    if scale is not None:
        y = y * torch.from_numpy(scale).double().view(1, -1, 1, 1)
    if shift is not None:
        y = y + torch.from_numpy(shift).double().view(1, -1, 1, 1)
    y = y.permute(0, 2, 3, 1)
    if res is not None:
        y = y + torch.from_numpy(res).double()
    if relu:
        y = y.clamp_min(0)
    return y.numpy()


CASES = [
    # B, H, W, Cin, Cout, ks, stride, res, relu
    (2, 56, 56, 32, 32, 3, 1, True, True),     # HRNet-W32 branch 0 BasicBlock conv2
    (3, 28, 28, 64, 64, 3, 1, False, True),
    (5, 14, 14, 128, 128, 3, 1, True, True),
    (17, 7, 7, 256, 256, 3, 1, False, False),
    (2, 56, 56, 48, 48, 3, 1, True, True),     # HRNet-W48
    (3, 14, 14, 192, 192, 3, 1, False, True),
    (2, 56, 56, 64, 256, 1, 1, True, True),    # Bottleneck expand
    (2, 56, 56, 256, 64, 1, 1, False, True),
    (2, 112, 112, 64, 64, 3, 2, False, True),  # stem conv2
    (3, 56, 56, 32, 64, 3, 2, False, False),   # fuse down path
    (3, 28, 28, 96, 192, 3, 2, False, True),
    (4, 7, 7, 1024, 2048, 1, 1, False, True),  # hrnet_cls final layer
    (2, 56, 56, 256, 512, 1, 2, False, False), # resnet downsample 1x1 s2
    (1, 56, 56, 480, 128, 3, 1, False, True),  # PARE head first conv
    (1, 13, 9, 16, 16, 3, 1, True, True),      # ragged odd plane
    (1, 1, 1, 16, 16, 3, 1, False, False),     # degenerate plane
    (2, 15, 15, 32, 48, 3, 2, False, True),    # odd size stride 2
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_parity(case, cuda):
    from poco_amd import ops
    B, H, W, Cin, Cout, ks, stride, use_res, relu = case
    rng = np.random.default_rng(hash(case) % (2**32))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, ks, ks)) / np.sqrt(Cin * ks * ks)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    pad = (ks - 1) // 2
    Ho = (H + 2 * pad - ks) // stride + 1
    Wo = (W + 2 * pad - ks) // stride + 1
    res = rng.standard_normal((B, Ho, Wo, Cout)).astype(np.float32) if use_res else None
    ref = _ref(x, w, scale, shift, stride, res, relu)
    xd = torch.from_numpy(x).to(cuda)
    rd = torch.from_numpy(res).to(cuda) if use_res else None
    out = ops.conv2d_nhwc(xd, w, scale, shift, stride, rd, relu).cpu().numpy()
    assert out.shape == ref.shape
    err = np.abs(out - ref).max()
    assert err <= 2e-5 * max(1.0, np.abs(ref).max()), err


TILES = [(4, 2, 2, 1, 2, 1), (7, 2, 4, 1, 8, 1), (7, 1, 1, 2, 2, 1), (13, 2, 1, 1, 3, 1), (7, 2, 2, 2, 4, 1),
         (4, 1, 2, 4, 1, 2), (4, 1, 2, 3, 1, 2), (7, 1, 1, 6, 2, 1)]


@pytest.mark.parametrize("alg", [1, 2])
@pytest.mark.parametrize("case", [c for c in CASES if c[3] % 16 == 0][:15], ids=lambda c: "x".join(map(str, c)))
def test_conv_parity_lds_dma(case, alg, cuda):
    """ALG 1 (LDS-DMA double-buffered patch + weights) and ALG 2 (the same, persistent over tiles)."""
    from poco_amd import ops
    B, H, W, Cin, Cout, ks, stride, use_res, relu = case
    rng = np.random.default_rng(hash(case) % (2**32))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, ks, ks)) / np.sqrt(Cin * ks * ks)).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    pad = (ks - 1) // 2
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    res = rng.standard_normal((B, Ho, Wo, Cout)).astype(np.float32) if use_res else None
    ref = _ref(x, w, None, shift, stride, res, relu)
    # small generic tile: 4 sub-tiles x 1 n-tile per wave, 2x2 waves, R rows so that the block fits
    R = max(1, min(Ho, (2 * 4 * 16) // Wo))
    NI = max(1, (2 * 4 * 16) // (R * Wo)) if R == Ho else 1
    cfg = (4, 1, 2, 2, R, min(NI, B), alg)
    out = ops.conv2d_nhwc(torch.from_numpy(x).to(cuda), w, None, shift, stride,
                          torch.from_numpy(res).to(cuda) if use_res else None, relu, cfg=cfg).cpu().numpy()
    assert np.abs(out - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("alg,nt", [(3, 1), (3, 2), (4, 1), (4, 2), (4, 3)])
@pytest.mark.parametrize("case", [c for c in CASES if c[5] == 3 and c[6] == 1 and c[3] % 16 == 0] +
                         [(3, 7, 7, 32, 32, 3, 1, True, True), (2, 28, 28, 96, 96, 3, 1, True, False)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv_parity_winograd(case, alg, nt, cuda):
    """ALG 3: Winograd F(2x2,3x3) on fp32 MFMA; odd planes (7x7, 13x9, 1x1) exercise the tile overhang."""
    from poco_amd import ops
    B, H, W, Cin, Cout, ks, stride, use_res, relu = case
    if (Cout // 16) % nt and alg == 3:
        pytest.skip("n-tiles not divisible")
    cap = 128 if alg == 3 else 64
    rng = np.random.default_rng(hash(case) % (2**32))
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    res = rng.standard_normal((B, H, W, Cout)).astype(np.float32) if use_res else None
    ref = _ref(x, w, scale, shift, 1, res, relu)
    TX, Hc = (W + 1) // 2, (H + 1) // 2 * 2
    R = 2
    while R + 2 <= Hc and ((R + 2) // 2) * TX <= cap:
        R += 2
    tiles = (R // 2) * TX
    WM = -(-tiles // 16)
    NI = min(B, max(1, (WM * 16) // tiles)) if R >= H else 1
    cfg = (1, nt, WM, 1 if alg == 3 else 2, R, NI, alg)
    out = ops.conv2d_nhwc(torch.from_numpy(x).to(cuda), w, scale, shift, 1,
                          torch.from_numpy(res).to(cuda) if use_res else None, relu, cfg=cfg).cpu().numpy()
    assert np.abs(out - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("nt", [1, 2, 3])
@pytest.mark.parametrize("case", [(1, 7, 7, 384, 384, True), (1, 14, 14, 192, 192, True), (16, 7, 7, 64, 48, False), (5, 14, 14, 16, 32, True),
                                  (2, 28, 28, 96, 96, True), (37, 56, 56, 32, 64, True), (3, 13, 9, 48, 16, False), (1, 14, 14, 32, 32, False)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv_winograd_half_deep_rings(case, nt, cuda):
    """ALG 4 with 3-deep raw / U rings (round 4, cfg.MT = 3): the fragments of slice c+2 and the patch of slice c+3 are requested at
    slice c and awaited with a counted s_waitcnt vmcnt that leaves one batch in flight (small-batch latency).  Same MFMAs in the same
    order as the 2-deep kernel: BITWISE its result, for 1 ... 24 slices (K = 16 ... 384), one tile per block and persistent blocks that
    walk several; configurations whose rings do not fit the LDS are refused."""
    from tests import util as _u
    if not _u.has_experiments():
        pytest.skip("experiment build only (python -m poco_amd.build --experiments; POCO_HIP_LIB=poco_amd/lib/exp/libpoco_hip_experiments.so)")
    from poco_amd import ops
    B, H, W, Cin, Cout, use_res = case
    rng = np.random.default_rng(B * 7 + Cin + Cout)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    res = rng.standard_normal((B, H, W, Cout)).astype(np.float32) if use_res else None
    TX, Hc = (W + 1) // 2, (H + 1) // 2 * 2
    R = 2
    while R + 2 <= Hc and ((R + 2) // 2) * TX <= 64:
        R += 2
    tiles = (R // 2) * TX
    WM = -(-tiles // 16)
    NI = min(B, max(1, (WM * 16) // tiles)) if R >= H else 1
    args = (torch.from_numpy(x).to(cuda), w, None, shift, 1, torch.from_numpy(res).to(cuda) if use_res else None, True)
    two = ops.conv2d_nhwc(*args, cfg=(1, nt, WM, 2, R, NI, 4)).cpu().numpy()
    plane = (NI * (R + 2) * (2 * TX + 2) + 63) // 64 * 64
    if (3 * 4 * plane + 3 * 16 * nt * 64) * 16 > 160 * 1024:
        with pytest.raises(RuntimeError):
            ops.conv2d_nhwc(*args, cfg=(3, nt, WM, 2, R, NI, 4))
        return
    three = ops.conv2d_nhwc(*args, cfg=(3, nt, WM, 2, R, NI, 4)).cpu().numpy()
    ref = _ref(x, w, None, shift, 1, res, True)
    assert np.abs(two - ref).max() <= 1e-4 * max(1.0, np.abs(ref).max())
    assert np.array_equal(two, three)


@pytest.mark.parametrize("cfg", TILES + [t + (1,) for t in TILES] + [t + (2,) for t in TILES])
def test_conv_explicit_tiles(cfg, cuda):
    """Every tile decomposition must give the same answer (asymmetric weights catch transposes)."""
    from poco_amd import ops
    rng = np.random.default_rng(7)
    B, H, W, Cin, Cout = 37, 56, 56, 32, 64     # enough tiles that persistent blocks walk several
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / 17.0).astype(np.float32)
    ref = _ref(x, w, None, None, 1, None, False)
    out = ops.conv2d_nhwc(torch.from_numpy(x).to(cuda), w, cfg=cfg).cpu().numpy()
    assert np.abs(out - ref).max() <= 2e-5 * np.abs(ref).max()


LINEAR = [(64, 2224, 1024), (64, 1024, 160), (37, 176, 1024), (1, 2048, 224), (100, 3296, 512), (129, 448, 32)]


@pytest.mark.parametrize("wm", [0, 1, 4, 8, 16])
@pytest.mark.parametrize("case", LINEAR, ids=lambda c: "x".join(map(str, c)))
def test_linear_small_m(case, wm, cuda):
    """ALG 5: Linear layers (cliff_head.py:104-118, poco_head.py:96-154) as B-row GEMMs with the K dimension
    split over the waves of a block; wm = 0 checks that the heuristic picks it for 1x1 planes."""
    from poco_amd import ops
    B, Cin, Cout = case
    rng = np.random.default_rng(B * 7 + Cin)
    x = rng.standard_normal((B, 1, 1, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    res = rng.standard_normal((B, 1, 1, Cout)).astype(np.float32)
    ref = _ref(x, w, None, shift, 1, res, True)
    cfg = None if wm == 0 else (1, 1, wm, 1, 1, 1, 5)
    out = ops.conv2d_nhwc(torch.from_numpy(x).to(cuda), w, None, shift, 1, torch.from_numpy(res).to(cuda), True,
                          cfg=cfg).cpu().numpy()
    assert np.abs(out - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("form", [1, 2, 3])       # cfg.R: 64 pixels per block x 4 K steps in flight (rounds 4-5) | 16 x 8 (round 6) | 32 x 6
@pytest.mark.parametrize("wm", [1, 4, 16])
@pytest.mark.parametrize("case", [(1, 7, 7, 384, 384, 1, True), (1, 14, 14, 192, 192, 1, True), (4, 28, 28, 96, 96, 1, False), (1, 56, 56, 48, 48, 1, True),
                                  (1, 56, 56, 48, 96, 2, False), (3, 28, 28, 96, 192, 2, True), (2, 14, 14, 192, 384, 2, False), (1, 13, 9, 32, 16, 1, True),
                                  (2, 15, 15, 32, 48, 2, False), (1, 1, 1, 16, 16, 1, False), (5, 7, 7, 16, 32, 2, True)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv3x3_split_k_small_batch(case, wm, form, cuda):
    """ALG 5 for 3x3 convs (round 4, csrc/linear_mfma.hip conv3x3_splitk_kernel): the direct conv as a GEMM over K = 9 Cin / 16 steps
    dealt to the wm waves of a block, one LDS reduction in a fixed order - the small-batch form of hrnet.py:42-58 (BasicBlock convs)
    and :208-236 (stride-2 fuse convs).  Exact fp32 fma chains: same tolerance as the direct kernels."""
    from poco_amd import ops
    B, H, W, Cin, Cout, stride, use_res = case
    rng = np.random.default_rng(B * 31 + Cin + Cout + stride)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(Cin * 9)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = rng.standard_normal((B, Ho, Wo, Cout)).astype(np.float32) if use_res else None
    ref = _ref(x, w, scale, shift, stride, res, True)
    out = ops.conv2d_nhwc(torch.from_numpy(x).to(cuda), w, scale, shift, stride, torch.from_numpy(res).to(cuda) if use_res else None,
                          True, cfg=(1, 1, wm, 1, form, 1, 5)).cpu().numpy()
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max()), np.abs(out - ref).max()


@pytest.mark.parametrize("wm", [1, 2, 4, 8])
@pytest.mark.parametrize("case", [(2, 56, 56, 64, 256), (3, 14, 14, 192, 144), (5, 7, 7, 384, 336), (1, 13, 9, 32, 16)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv1x1_split_k(case, wm, cuda):
    """ALG 5 on spatial planes: 1x1 stride-1 convs as a split-K GEMM over the L16 pixel rows (hrnet.py:196-207)."""
    from poco_amd import ops
    B, H, W, Cin, Cout = case
    rng = np.random.default_rng(B * 131 + Cin)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    res = rng.standard_normal((B, H, W, Cout)).astype(np.float32)
    ref = _ref(x, w, scale, shift, 1, res, True)
    out = ops.conv2d_nhwc(torch.from_numpy(x).to(cuda), w, scale, shift, 1, torch.from_numpy(res).to(cuda), True,
                          cfg=(1, 1, wm, 1, 1, 1, 5)).cpu().numpy()
    assert np.abs(out - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


GEMM1X1 = [
    # B, H, W, Cin, Cout, stride, res
    (2, 56, 56, 64, 256, 1, True),      # Bottleneck expand + residual (resnet.py:101-121)
    (2, 56, 56, 256, 64, 1, False),     # Bottleneck reduce
    (3, 14, 14, 1024, 256, 1, False),
    (3, 14, 14, 256, 1024, 1, True),
    (5, 7, 7, 2048, 512, 1, False),     # 49-pixel planes: sub-tiles straddle images
    (2, 56, 56, 256, 512, 2, False),    # downsample.0 (1x1 stride 2)
    (3, 14, 14, 1024, 2048, 2, False),
    (1, 13, 9, 32, 48, 1, True),        # ragged plane, Cout not a multiple of NT*16
    (2, 15, 11, 48, 16, 2, True),       # odd plane, stride 2
    (1, 1, 1, 16, 16, 1, False),        # one pixel
    (2, 28, 28, 80, 144, 1, True),      # K = 5 slices: exercises the depth-2 and depth-3 tails
]
GEMM1X1_CFG = [(2, 4, 2, 2, 2, 1, 6), (4, 2, 4, 2, 3, 1, 6), (4, 4, 1, 4, 2, 1, 6), (4, 4, 2, 2, 3, 1, 6), (7, 2, 2, 4, 3, 1, 6),
               (7, 4, 1, 2, 2, 1, 6), (7, 4, 2, 2, 3, 1, 6), (8, 2, 8, 1, 2, 1, 6), (8, 2, 1, 1, 3, 1, 6),
               # pinned load schedules (NI 2..6): early / late in the slice, both prefetch depths
               (7, 2, 2, 2, 2, 2, 6), (7, 2, 1, 1, 3, 3, 6), (7, 4, 1, 1, 2, 4, 6), (7, 4, 2, 2, 3, 5, 6), (4, 2, 1, 2, 2, 6, 6),
               (2, 4, 1, 1, 3, 5, 6), (8, 2, 2, 2, 2, 3, 6), (4, 4, 2, 1, 2, 4, 6),
               # ALG 9: the same GEMM with coalesced global traffic through a wave-private LDS transposition (csrc/gemm1x1t.hip)
               (7, 4, 1, 1, 1, 1, 9), (7, 4, 2, 2, 1, 1, 9), (7, 2, 2, 4, 1, 1, 9), (4, 4, 1, 4, 1, 1, 9), (8, 2, 8, 1, 1, 1, 9),
               (8, 2, 1, 1, 1, 1, 9)]


@pytest.mark.parametrize("cfg", GEMM1X1_CFG, ids=lambda c: "-".join(map(str, c)))
@pytest.mark.parametrize("case", GEMM1X1, ids=lambda c: "x".join(map(str, c)))
def test_conv1x1_register_gemm(case, cfg, cuda):
    """ALG 6: 1x1 convs (stride 1|2) as a register-direct GEMM without LDS (csrc/gemm1x1.hip); Bottleneck 1x1 convs of
    resnet.py:101-121 / hrnet.py:79-99 incl. the stride-2 downsample path."""
    from poco_amd import ops
    B, H, W, Cin, Cout, stride, has_res = case
    rng = np.random.default_rng(B * 977 + Cin + Cout)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = rng.standard_normal((B, Ho, Wo, Cout)).astype(np.float32) if has_res else None
    ref = _ref(x, w, scale, shift, stride, res, True)
    out = ops.conv2d_nhwc(torch.from_numpy(x).to(cuda), w, scale, shift, stride,
                          None if res is None else torch.from_numpy(res).to(cuda), True, cfg=cfg).cpu().numpy()
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("cfg", [(4, 4, 2, 2, 2, 1, 12), (4, 2, 1, 4, 2, 1, 12), (2, 4, 4, 1, 2, 1, 12), (2, 2, 1, 1, 2, 1, 12),
                                 (4, 4, 2, 2, 8, 1, 12)],      # R = 8: the LDS-tiled kernel
                         ids=lambda c: "-".join(map(str, c)))
@pytest.mark.parametrize("case", [(2, 56, 56, 64, 256, 1, True), (3, 14, 14, 1024, 512, 1, False), (2, 28, 28, 512, 128, 1, True),
                                  (1, 13, 9, 32, 16, 1, True), (2, 56, 56, 256, 512, 2, False), (5, 7, 7, 96, 48, 1, False)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv1x1_split_f16_experiment(case, cfg, cuda):
    """ALG 12 (EXPERIMENT, csrc/gemm1x1h.hip): 1x1 convs with every operand split into fp16 hi + lo, three
    v_mfma_f32_16x16x32_f16 per product, fp32 accumulation.  22 mantissa bits per operand: within 2e-5 of the fp64 conv like
    the fp32 kernels (values O(1), K up to 1024, odd K = 3 slices of 32)."""
    from tests import util as _u
    if not _u.has_experiments():
        pytest.skip("experiment build only (python -m poco_amd.build --experiments; POCO_HIP_LIB=poco_amd/lib/exp/libpoco_hip_experiments.so)")
    from poco_amd import ops
    B, H, W, Cin, Cout, stride, has_res = case
    rng = np.random.default_rng(B * 977 + Cin + Cout)
    x = (rng.standard_normal((B, H, W, Cin)) * rng.choice([1e-3, 1.0, 30.0], (1, 1, 1, Cin))).astype(np.float32)   # small / O(1) / large channels
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = rng.standard_normal((B, Ho, Wo, Cout)).astype(np.float32) if has_res else None
    ref = _ref(x, w, scale, shift, stride, res, True)
    out = ops.conv2d_nhwc(torch.from_numpy(x).to(cuda), w, scale, shift, stride,
                          None if res is None else torch.from_numpy(res).to(cuda), True, cfg=cfg).cpu().numpy()
    err = np.abs(out - ref).max() / max(1.0, np.abs(ref).max())
    print("split-f16 relative deviation %.2e" % err)
    assert err <= 2e-5


GEMM1X1_SK_CFG = [(7, 4, 2, 1, 2, 1, 14), (7, 4, 4, 1, 2, 6, 14), (7, 2, 8, 1, 2, 3, 14), (7, 2, 4, 1, 3, 6, 14), (4, 4, 4, 1, 3, 6, 14),
                  (4, 4, 8, 1, 2, 1, 14), (4, 2, 8, 1, 3, 3, 14), (2, 4, 8, 1, 3, 6, 14), (4, 4, 1, 1, 3, 6, 14)]


@pytest.mark.parametrize("cfg", GEMM1X1_SK_CFG, ids=lambda c: "-".join(map(str, c)))
@pytest.mark.parametrize("case", [c for c in GEMM1X1 if c[5] == 1] + [(64, 14, 14, 1024, 256, 1, True), (7, 7, 7, 2048, 512, 1, False)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv1x1_stream_k_gemm(case, cfg, cuda):
    """ALG 14 (csrc/gemm1x1sk.hip): the 1x1 stride-1 GEMM with the (tile, K slice) units dealt evenly to a persistent grid of
    waves; tiles that straddle waves are finished from partial accumulators exchanged through a scratch buffer (flags, agent-scope
    accesses).  Against the fp64 conv like ALG 6 / 9; cases from one tile for the whole grid (every wave a slice of it) to the
    bench-size 14x14 1024->256 of 64 crops (1.75 tiles per wave), with and without residual; run twice: the flags are re-armed."""
    from poco_amd import ops
    B, H, W, Cin, Cout, stride, has_res = case
    rng = np.random.default_rng(B * 977 + Cin + Cout)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 1, 1)) / np.sqrt(Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    res = rng.standard_normal((B, H, W, Cout)).astype(np.float32) if has_res else None
    ref = _ref(x, w, scale, shift, 1, res, True)
    xd = torch.from_numpy(x).to(cuda)
    rd = None if res is None else torch.from_numpy(res).to(cuda)
    out = ops.conv2d_nhwc(xd, w, scale, shift, 1, rd, True, cfg=cfg)
    again = ops.conv2d_nhwc(xd, w, scale, shift, 1, rd, True, cfg=cfg)
    assert torch.equal(out, again)                      # deterministic split, flags lowered by the finishing waves
    assert np.abs(out.cpu().numpy() - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_conv1x1_register_gemm_rejects_3x3(cuda):
    from poco_amd import ops
    x = torch.zeros(1, 8, 8, 16, device=cuda)
    w = np.zeros((16, 16, 3, 3), np.float32)
    with pytest.raises(RuntimeError):
        ops.conv2d_nhwc(x, w, cfg=(4, 2, 2, 2, 2, 1, 6))


GEMM3X3 = [
    # B, H, W, Cin, Cout, stride, residual
    (2, 56, 56, 48, 144, 2, False),     # HRNet-W48 merged fuse down-path conv (9 n-tiles)
    (3, 28, 28, 96, 192, 2, True),
    (2, 14, 14, 192, 384, 2, False),
    (1, 14, 14, 512, 1024, 2, False),   # cls head: K = 288 steps
    (3, 13, 9, 32, 48, 2, True),        # ragged plane, odd sizes: the last row / column taps fall outside
    (2, 7, 7, 64, 16, 2, False),        # 7 -> 4
    (1, 1, 1, 16, 16, 2, False),        # one pixel: eight of nine taps are padding
    (2, 12, 10, 48, 80, 1, True),       # stride 1 through the same kernel
    (5, 56, 56, 16, 32, 2, True),       # many sub-tiles per image, K = 9 steps
]
GEMM3X3_CFG = [(2, 4, 2, 2, 2, 1, 10), (4, 2, 4, 2, 3, 1, 10), (4, 3, 1, 1, 2, 1, 10), (4, 3, 2, 3, 3, 3, 10), (4, 4, 1, 4, 2, 6, 10),
               (7, 2, 2, 4, 3, 1, 10), (7, 3, 1, 2, 2, 3, 10), (7, 4, 1, 2, 2, 1, 10), (7, 4, 2, 2, 3, 6, 10), (8, 2, 8, 1, 2, 1, 10),
               (8, 2, 1, 1, 3, 3, 10)]


@pytest.mark.parametrize("cfg", GEMM3X3_CFG, ids=lambda c: "-".join(map(str, c)))
@pytest.mark.parametrize("case", GEMM3X3, ids=lambda c: "x".join(map(str, c)))
def test_conv3x3_register_gemm(case, cfg, cuda):
    """ALG 10 (round 3): 3x3 convs (stride 1|2, pad 1) as a register-direct gather GEMM over K = 9*Cin without LDS
    (csrc/gemm3x3.hip): the stride-2 convs of hrnet.py:196-264 (fuse down paths, transitions), hrnet_cls.py:306-353
    (cls head) and resnet.py:101-121 (layer2-4.0 conv2), against the fp64 convolution incl. BN scale / shift, residual, ReLU."""
    from poco_amd import ops
    B, H, W, Cin, Cout, stride, has_res = case
    rng = np.random.default_rng(B * 977 + Cin + Cout + H)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    res = rng.standard_normal((B, Ho, Wo, Cout)).astype(np.float32) if has_res else None
    ref = _ref(x, w, scale, shift, stride, res, True)
    out = ops.conv2d_nhwc(torch.from_numpy(x).to(cuda), w, scale, shift, stride,
                          None if res is None else torch.from_numpy(res).to(cuda), True, cfg=cfg).cpu().numpy()
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())


def test_conv3x3_register_gemm_rejects_1x1(cuda):
    from poco_amd import ops
    x = torch.zeros(1, 8, 8, 16, device=cuda)
    with pytest.raises(RuntimeError):
        ops.conv2d_nhwc(x, np.zeros((16, 16, 1, 1), np.float32), cfg=(4, 2, 2, 2, 2, 1, 10))


WINO4G = [
    # B, H, W, Cin, Cout, residual, relu
    (64, 7, 7, 384, 384, True, True),     # HRNet-W48 branch 3 BasicBlock conv2 at the bench batch size
    (5, 7, 7, 256, 256, False, True),     # W32
    (3, 8, 8, 32, 48, True, False),       # exact 2 x 2 tiles
    (7, 5, 3, 16, 32, True, True),        # ragged plane: 2 x 1 tiles, clipped rows and columns
    (2, 1, 2, 16, 16, False, False),      # one tile per image
    (9, 4, 4, 48, 16, True, True),        # one tile, T = 9 (padded to 128 tile slots)
    (33, 6, 7, 80, 112, False, True),     # T = 132 > one 128-tile group, Cin / Cout no multiple of the n-tile groups
]
WINO4G_CFG = [(4, 4, 4, 1, 3, 1, 11), (4, 4, 1, 1, 2, 1, 11), (4, 2, 2, 2, 3, 1, 11), (2, 4, 1, 4, 2, 1, 11), (8, 2, 1, 1, 3, 1, 11),
              (8, 2, 2, 4, 2, 1, 11), (2, 4, 8, 1, 3, 1, 11)]


@pytest.mark.parametrize("cfg", WINO4G_CFG, ids=lambda c: "-".join(map(str, c)))
@pytest.mark.parametrize("case", WINO4G, ids=lambda c: "x".join(map(str, c)))
def test_conv_winograd_f4x4_as_gemm(case, cfg, cuda):
    """ALG 11 (round 3): Winograd F(4x4,3x3) on small planes as input transform -> 36 position GEMMs -> output transform
    (csrc/conv_wino4g.hip; the 7x7 BasicBlock convs of hrnet.py:42-58) against the fp64 convolution."""
    from poco_amd import ops
    B, H, W, Cin, Cout, has_res, relu = case
    rng = np.random.default_rng(B * 131 + Cin + Cout + H)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    res = rng.standard_normal((B, H, W, Cout)).astype(np.float32) if has_res else None
    ref = _ref(x, w, scale, shift, 1, res, relu)
    out = ops.conv2d_nhwc(torch.from_numpy(x).to(cuda), w, scale, shift, 1, None if res is None else torch.from_numpy(res).to(cuda),
                          relu, cfg=cfg).cpu().numpy()
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), np.abs(out - ref).max()


def test_conv_winograd_f4x4_as_gemm_rejects_large_planes(cuda):
    from poco_amd import ops
    with pytest.raises(RuntimeError):
        ops.conv2d_nhwc(torch.zeros(1, 28, 28, 16, device=cuda), np.zeros((16, 16, 3, 3), np.float32), cfg=(4, 4, 1, 1, 2, 1, 11))


WINO4 = [
    # B, H, W, Cin, Cout, res
    (2, 56, 56, 48, 48, True),      # HRNet-W48 branch 0 BasicBlock conv
    (3, 28, 28, 96, 96, False),
    (2, 56, 56, 32, 32, True),      # W32
    (5, 14, 14, 64, 48, True),      # 14 = 3.5 tiles: ragged right/bottom tiles
    (3, 7, 7, 32, 16, False),
    (1, 13, 9, 16, 32, True),       # ragged plane
    (1, 1, 1, 16, 16, False),       # one pixel
    (7, 12, 8, 160, 80, True),      # K = 10 slices; tiles not a multiple of the 16 / 32 per block
    (40, 56, 56, 16, 48, True),     # 280 work items at NT = 3 (840 at NT = 1) > 256 blocks: persistent blocks walk several items
]


def _wino4_cfg(H, W, nt, alg=7):
    """(R, NI) of ALG 7 / 8 for a plane: as many 4-row tile bands as fit 32 tiles, whole images when several fit."""
    TX, Hc = (W + 3) // 4, (H + 3) // 4 * 4
    R = Hc if (Hc // 4) * TX <= 32 else max(4, 32 // TX * 4)
    NI = max(1, min(32 // ((R // 4) * TX), 1024 // ((R + 2) * (4 * TX + 2)))) if R == Hc else 1
    npos = lambda ni: ni * (R + 2) * (4 * TX + 2)
    lds = lambda ni: 4 * ((npos(ni) + npos(ni) // 8 + 1 + 63) // 64 * 64 + 9 * nt * 64) * 16     # 2 buffers x 2 slices, skewed slots
    while NI > 1 and (lds(NI) > 160 * 1024 or (npos(NI) + npos(NI) // 8 + 1 + 63) // 64 * 64 > 1024):
        NI -= 1
    return (1, nt, 2, 4, R, NI, alg)


@pytest.mark.parametrize("alg", [7, 8])
@pytest.mark.parametrize("nt", [1, 2, 3])
@pytest.mark.parametrize("case", WINO4, ids=lambda c: "x".join(map(str, c)))
def test_conv_winograd_f4x4(case, nt, alg, cuda):
    """ALG 7 / ALG 8 (specialised waves: 8 MFMA waves + 4 producer waves, V staged in LDS): Winograd F(4x4,3x3) for the
    3x3 stride-1 convs of hrnet.py:42-58.
    Tolerance 2e-4 * max|ref|: the F(4x4) transforms (constants up to 8) amplify fp32 rounding ~10x over F(2x2)."""
    from poco_amd import ops
    B, H, W, Cin, Cout, has_res = case
    cfg = _wino4_cfg(H, W, nt, alg)
    rng = np.random.default_rng(B * 313 + Cin + Cout)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    res = rng.standard_normal((B, H, W, Cout)).astype(np.float32) if has_res else None
    ref = _ref(x, w, scale, shift, 1, res, True)
    out = ops.conv2d_nhwc(torch.from_numpy(x).to(cuda), w, scale, shift, 1,
                          None if res is None else torch.from_numpy(res).to(cuda), True, cfg=cfg).cpu().numpy()
    assert np.abs(out - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("nt", [1, 2, 3])
@pytest.mark.parametrize("case", WINO4 + [(64, 28, 28, 16, 32, True), (9, 56, 56, 16, 16, False), (33, 14, 14, 32, 32, True)],
                         ids=lambda c: "x".join(map(str, c)))
def test_conv_winograd_f4x4_flat_items(case, nt, cuda):
    """ALG 8 with FLAT items (round 4, cfg R = 4, NI = 0): a block's 32 MFMA tile columns carry 32 CONSECUTIVE tiles of the flattened
    (image, tile row, tile column) order instead of a rectangle of 2 x 14 / 4 x 7 tiles, the patch is a 6-row strip of tile-row
    fragments.  Items start anywhere in a tile row, span up to nine fragments and two images, the last item is partial; planes whose
    strip does not fit the LDS next to the U ring at this NT must be refused, everything else must equal the fp64 conv."""
    from poco_amd import ops
    B, H, W, Cin, Cout, has_res = case
    cfg = (1, nt, 2, 4, 4, 0, 8)
    TX = (W + 3) // 4
    fmax = (TX - 1 + 32 + TX - 1) // TX
    npos = 6 * (128 + 2 * fmax)
    raw = (npos + npos // 16 + 1 + 63) // 64 * 64
    fits = raw <= 1024 and (3 * raw + 3 * nt * 576 + 4 * 576) * 16 <= 160 * 1024
    rng = np.random.default_rng(B * 131 + Cin + Cout + nt)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    res = rng.standard_normal((B, H, W, Cout)).astype(np.float32) if has_res else None
    args = (torch.from_numpy(x).to(cuda), w, scale, shift, 1, None if res is None else torch.from_numpy(res).to(cuda), True)
    if not fits:
        with pytest.raises(RuntimeError):
            ops.conv2d_nhwc(*args, cfg=cfg)
        return
    out = ops.conv2d_nhwc(*args, cfg=cfg).cpu().numpy()
    ref = _ref(x, w, scale, shift, 1, res, True)
    assert np.abs(out - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), np.abs(out - ref).max()
    # bitwise the rectangular-item result: same arithmetic per tile, only the assignment of tiles to blocks differs
    rect = ops.conv2d_nhwc(*args, cfg=_wino4_cfg(H, W, nt, 8)).cpu().numpy()
    assert np.array_equal(out, rect)


@pytest.mark.parametrize("flat", [0, 1])
@pytest.mark.parametrize("nt", [1, 2, 3])
@pytest.mark.parametrize("case", WINO4 + [(64, 28, 28, 16, 32, True), (9, 56, 56, 16, 16, False), (33, 14, 14, 32, 32, True),
                                          (64, 14, 14, 48, 112, True), (3, 56, 56, 64, 128, True),
                                          (5, 14, 14, 64, 112, True)],      # U stream > activations: items walked strip-innermost (round 6)
                         ids=lambda c: "x".join(map(str, c)))
def test_conv_winograd_f4x4_whole_position_waves(case, nt, flat, cuda):
    """ALG 13 (round 5, conv_wino4w.hip): every MFMA wave owns all 36 positions of a 16-tile group for one n-tile, the output
    transform is register-only (no exchange rounds), the slice pipeline runs on across item boundaries (ring phases, the producers'
    windows and the padding lanes of the raw ring switch items mid-stream) and items are walked n-group-innermost.  Rectangular and
    flat items, one to many items per block, several n-groups with a partly empty last one (112 = 7 n-tiles), ragged planes: all
    must equal the fp64 conv; and since the arithmetic per (tile, channel) is ALG 8's up to the summation tree of the output
    transform, the result must also agree with ALG 8 to a few ulp."""
    from poco_amd import ops
    B, H, W, Cin, Cout, has_res = case
    rect = _wino4_cfg(H, W, nt, 8)
    cfg = (1, nt, 2, 1, 4, 0, 13) if flat else rect[:3] + (1,) + rect[4:6] + (13,)
    TX = (W + 3) // 4
    fmax = (TX - 1 + 32 + TX - 1) // TX
    npos = 6 * (128 + 2 * fmax)
    raw = (npos + npos // 16 + 1 + 63) // 64 * 64
    fits = not flat or raw <= 1024     # raw ring + V double buffer always fit (round 6: U goes straight into registers, no U ring in LDS)
    rng = np.random.default_rng(B * 131 + Cin + Cout + nt)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    res = rng.standard_normal((B, H, W, Cout)).astype(np.float32) if has_res else None
    args = (torch.from_numpy(x).to(cuda), w, scale, shift, 1, None if res is None else torch.from_numpy(res).to(cuda), True)
    if not fits:
        with pytest.raises(RuntimeError):
            ops.conv2d_nhwc(*args, cfg=cfg)
        return
    out = ops.conv2d_nhwc(*args, cfg=cfg).cpu().numpy()
    ref = _ref(x, w, scale, shift, 1, res, True)
    assert np.abs(out - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), np.abs(out - ref).max()
    a8 = ops.conv2d_nhwc(*args, cfg=rect).cpu().numpy()
    assert np.abs(out - a8).max() <= 2e-5 * max(1.0, np.abs(ref).max()), np.abs(out - a8).max()
    # twice the same launch: bitwise (no race between the item-crossing pipeline stages)
    assert np.array_equal(out, ops.conv2d_nhwc(*args, cfg=cfg).cpu().numpy())


@pytest.mark.parametrize("ms", [2, 4, 8])
@pytest.mark.parametrize("nt", [1, 3])
@pytest.mark.parametrize("case", [(64, 14, 14, 32, 48, True), (33, 14, 14, 16, 16, False), (5, 7, 7, 32, 16, True), (19, 13, 9, 16, 32, True),
                                  (3, 28, 28, 16, 16, True), (1, 1, 1, 16, 16, False), (16, 14, 14, 64, 64, True)],
                         ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("alg", [8, 13])
def test_conv_winograd_f4x4_mosaic_items(case, nt, ms, alg, cuda):
    """ALG 8 flat items over a MOSAIC (cfg R = 4 MS, NI = 0): MS x MS images share their one-pixel zero borders in one virtual plane
    (14 x 14 planes: 15 x 15 tiles per 4 x 4 images instead of 16 x 16), tiles straddle images and border lines, the last mosaic is
    partly empty when B is no multiple of MS^2.  Equal to the fp64 conv; pixels whose tile lies inside one image see the same
    arithmetic as with plain flat items, so the result differs from them by rounding only where the 4 x 4 tiling shifted."""
    from poco_amd import ops
    B, H, W, Cin, Cout, has_res = case
    cfg = (1, nt, 2, 4, 4 * ms, 0, 8) if alg == 8 else (1, nt, 2, 1, 4 * ms, 0, 13)      # (round 6: ALG 13 walks mosaics too)
    TX = (ms * (W + 1) - 1 + 3) // 4
    fmax = (TX - 1 + 32 + TX - 1) // TX
    npos = 6 * (128 + 2 * fmax)
    raw = (npos + npos // 16 + 1 + 63) // 64 * 64
    fits = raw <= 1024 and (alg == 13 or (3 * raw + 3 * nt * 576 + 4 * 576) * 16 <= 160 * 1024)
    rng = np.random.default_rng(B * 17 + Cin + Cout + nt + ms)
    x = rng.standard_normal((B, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    scale = rng.uniform(0.5, 1.5, Cout).astype(np.float32)
    shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
    res = rng.standard_normal((B, H, W, Cout)).astype(np.float32) if has_res else None
    args = (torch.from_numpy(x).to(cuda), w, scale, shift, 1, None if res is None else torch.from_numpy(res).to(cuda), True)
    if not fits:
        with pytest.raises(RuntimeError):
            ops.conv2d_nhwc(*args, cfg=cfg)
        return
    out = ops.conv2d_nhwc(*args, cfg=cfg).cpu().numpy()
    ref = _ref(x, w, scale, shift, 1, res, True)
    assert np.abs(out - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max()), np.abs(out - ref).max()


# ------------------------------------------------------------------------------------------------------------
# Table-driven: every (shape, cfg) pair the engine actually runs at the bench batch sizes
# ------------------------------------------------------------------------------------------------------------
def _tuned_entries(batches=(32, 64, 128)):
    import re
    from poco_amd import tune
    out = []
    for key, cfg in sorted(tune.load_table().items()):
        B, H, W, Cin, Cout, ks, stride = map(int, re.fullmatch(r"(\d+)x(\d+)x(\d+)x(\d+)x(\d+)k(\d+)s(\d+)", key).groups())
        if B in batches and cfg and cfg[0] > 0:
            out.append((key, (B, H, W, Cin, Cout, ks, stride), tuple(cfg)))
    return out


def _conv_fp64_gpu(x, w, shift, stride, res, relu):
    """fp64 conv on the GPU from fp64 matmuls only (no vendor conv library): x NHWC fp32 cuda, w OIHW numpy."""
    B, H, W, Cin = x.shape
    Cout, _, ks, _ = w.shape
    pad = (ks - 1) // 2
    Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
    xp = F.pad(x.double(), (0, 0, pad, pad, pad, pad))
    wd = torch.from_numpy(w).to(x.device).double()
    y = torch.zeros(B * Ho * Wo, Cout, device=x.device, dtype=torch.float64)
    for r in range(ks):
        for s in range(ks):
            v = xp[:, r:r + (Ho - 1) * stride + 1:stride, s:s + (Wo - 1) * stride + 1:stride, :]
            y.addmm_(v.reshape(-1, Cin), wd[:, :, r, s].t())
    y = y.view(B, Ho, Wo, Cout) + torch.from_numpy(shift).to(x.device).double()
    if res is not None:
        y = y + res.double()
    return y.clamp_min(0) if relu else y


ALG_TOL = {3: 1e-4, 4: 1e-4, 7: 2e-4, 8: 2e-4, 11: 2e-4}     # Winograd F(2x2): transforms amplify fp32 rounding ~5x, F(4x4) ~10x


@pytest.mark.parametrize("batch", [1, 4, 16, 32, 64, 128])
def test_tuned_table_entries(batch, cuda):
    """EVERY distinct (shape, cfg) of poco_amd/tuned/gfx950.json at the bench batch sizes goes
    through poco_op_conv2d at that batch size against an fp64 conv (all crops, all pixels), so a wrong tile in a tuned
    Winograd / persistent / LDS-DMA entry cannot hide behind an insensitive whole-model output.  An entry the
    library refuses for its shape fails the test (at run time the engine would silently fall back to its heuristic and lose
    the published throughput)."""
    from poco_amd import ops
    from poco_amd._lib import PocoHipError
    entries = _tuned_entries((batch,))
    assert entries, "no tuned entries for this batch size"
    gen = torch.Generator(device=cuda)
    worst, refused, by_alg = {}, [], {}
    for key, (B, H, W, Cin, Cout, ks, stride), cfg in entries:
        gen.manual_seed(zlib.crc32(key.encode()) % (2 ** 31))
        x = torch.randn((B, H, W, Cin), device=cuda, generator=gen)
        rng = np.random.default_rng(zlib.crc32(key.encode()))
        w = (rng.standard_normal((Cout, Cin, ks, ks)) / np.sqrt(Cin * ks * ks)).astype(np.float32)
        shift = rng.uniform(-0.5, 0.5, Cout).astype(np.float32)
        pad = (ks - 1) // 2
        Ho, Wo = (H + 2 * pad - ks) // stride + 1, (W + 2 * pad - ks) // stride + 1
        res = torch.randn((B, Ho, Wo, Cout), device=cuda, generator=gen)
        try:
            out = ops.conv2d_nhwc(x, w, None, shift, stride, res, True, cfg=cfg)
        except (PocoHipError, RuntimeError) as e:
            refused.append((key, cfg, str(e)[:80]))
            continue
        ref = _conv_fp64_gpu(x, w, shift, stride, res, True)
        err = float((out.double() - ref).abs().max())
        tol = ALG_TOL.get(cfg[6], 2e-5) * max(1.0, float(ref.abs().max()))
        by_alg[cfg[6]] = max(by_alg.get(cfg[6], 0.0), err / max(1.0, float(ref.abs().max())))
        if err > tol:
            worst[key] = (cfg, err, tol)
        del x, res, out, ref
    print(f"B={batch}: {len(entries)} tuned entries, {len(refused)} refused by the library; worst relative deviation per ALG:",
          {a: "%.1e" % v for a, v in sorted(by_alg.items())})
    for r in refused:
        print("  refused:", r)
    assert not worst, worst
    assert refused == [], refused        # VERDICT r2 weak #2 / ADVICE r2: every entry of the committed table must be accepted
