"""Stand-alone HIP head operators (C ABI: poco_op_part_attention / poco_op_lc2d_pose / poco_op_rot6d) against vectors made by the
REFERENCE's own modules (tests/golden/ops.npz <- oracle/gen_golden.py: KeypointAttention, LocallyConnected2d, rot6d_to_rotmat) and,
at the PARE head's real sizes, against the oracle (which test_oracle_golden.py pins to the same vectors).  VERDICT r1 rows a6 / a7
were covered end to end only."""
from pathlib import Path

import numpy as np
import pytest
import torch

from poco_amd import ops

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden" / "ops.npz"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.fixture(scope="module")
def cuda():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def test_part_attention_golden(gold, cuda):
    """layers/keypoint_attention.py:34-48 on the reference-made vector (2 crops, 20 feature channels, 24 parts, 9x7 pixels:
    odd sizes on purpose - padded channels, a pixel count that no split divides)."""
    out = ops.part_attention(torch.from_numpy(gold["ka_feat"]).to(cuda), torch.from_numpy(gold["ka_heat"]).to(cuda))
    torch.cuda.synchronize()
    assert out.shape == (2, 20, 24)
    assert np.abs(out.cpu().numpy() - gold["ka_out"]).max() < 1e-5


@pytest.mark.parametrize("B,C,H,W", [(3, 128, 56, 56), (1, 64, 56, 56), (5, 128, 7, 7)])
def test_part_attention_pare_sizes(B, C, H, W, cuda):
    """The PARE head's sizes (pare_head.py:794-796: 128- and 64-channel feature maps at 56x56) with peaked heat maps (softmax over
    3136 pixels with logits up to +-12: the split-pixel online softmax has to combine very different local maxima)."""
    from oracle import poco_ref
    r = np.random.default_rng(B * 1000 + C)
    feat = torch.from_numpy(r.standard_normal((B, C, H, W)).astype(np.float32))
    heat = torch.from_numpy((4.0 * r.standard_normal((B, 24, H, W))).astype(np.float32))
    heat[:, :, H // 2, W // 3] += 9.0
    ref = poco_ref.keypoint_attention(feat, heat).numpy()
    out = ops.part_attention(feat.to(cuda), heat.to(cuda)).cpu().numpy()
    assert np.abs(out - ref).max() < 2e-5 * max(1.0, np.abs(ref).max())


def test_lc2d_golden(gold, cuda):
    """layers/locallyconnected2d.py:27-37 (128 -> 6 per joint) on the reference-made vector."""
    x = torch.from_numpy(gold["lc_x"][..., 0]).to(cuda)                   # [3,128,24]
    w = torch.from_numpy(gold["lc_w"][0, :, :, :, 0, 0]).to(cuda)         # [6,128,24]
    out = ops.lc2d_pose(x, w)
    torch.cuda.synchronize()
    ref = gold["lc_out"][..., 0].transpose(0, 2, 1)                        # [3,6,24] -> [3,24,6]
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-5


def test_rot6d_golden(gold, cuda):
    """utils/geometry.py:247-261 on the reference-made vector (48 = 2 x 24 joints)."""
    x = torch.from_numpy(gold["rot6d_in"]).reshape(2, 24, 6).to(cuda)
    out = ops.rot6d(x).cpu().numpy().reshape(48, 3, 3)
    assert np.abs(out - gold["rot6d_out"]).max() < 1e-6
    R = out.astype(np.float64)
    assert np.abs(R.transpose(0, 2, 1) @ R - np.eye(3)).max() < 1e-5


def test_head_ops_reject_bad_arguments(cuda):
    from poco_amd._lib import PocoHipError
    with pytest.raises(PocoHipError):
        import ctypes as C
        from poco_amd._lib import check, lib
        check(lib().poco_op_part_attention(C.c_void_p(0), 32, C.c_void_p(0), 16, 1, 4, 4, C.c_void_p(0), C.c_void_p(0)), "part_attention")
