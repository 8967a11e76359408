"""Python wrappers of the stand-alone HIP operators (thin: pointers in, pointers out)."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import check, current_stream, fptr, lib


def _cfg(cfg):
    if cfg is None:
        return C.c_void_p(0), None
    arr = (C.c_int * 7)(*(tuple(cfg) + (0,) * (7 - len(cfg))))
    return C.cast(arr, C.c_void_p), arr


def to_l16(x: torch.Tensor) -> torch.Tensor:
    """NHWC [B,H,W,C] (C % 16 == 0) -> the library's activation layout [B,H,C/16,W,16] (csrc/common.h)."""
    B, H, W, C = x.shape
    assert C % 16 == 0
    return x.view(B, H, W, C // 16, 16).permute(0, 1, 3, 2, 4).contiguous()


def from_l16(y: torch.Tensor) -> torch.Tensor:
    """[B,H,C/16,W,16] -> NHWC [B,H,W,C]."""
    B, H, C16, W, _ = y.shape
    return y.permute(0, 1, 3, 2, 4).reshape(B, H, W, C16 * 16).contiguous()


def conv2d_nhwc(x: torch.Tensor, weight: np.ndarray, scale=None, shift=None, stride=1, residual=None,
                relu=False, cfg=None) -> torch.Tensor:
    """x [B,H,W,Cin] cuda fp32 NHWC; weight OIHW numpy fp32 (host). Returns NHWC output."""
    assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous()
    B, H, W, Cin = x.shape
    Cout, Cin2, ks, ks2 = weight.shape
    assert Cin2 == Cin and ks == ks2
    pad = (ks - 1) // 2
    Ho = (H + 2 * pad - ks) // stride + 1
    Wo = (W + 2 * pad - ks) // stride + 1
    # the operator works on the library's L16 layout; Cin/Cout are padded to multiples of 16 inside the op,
    # so NHWC tensors are converted here (tests / tuning only - the engine never leaves L16)
    assert Cin % 16 == 0 and Cout % 16 == 0, "conv2d_nhwc: channel counts must be multiples of 16"
    out = torch.empty((B, Ho, Cout // 16, Wo, 16), device=x.device, dtype=torch.float32)
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    scale = None if scale is None else np.ascontiguousarray(scale, dtype=np.float32)
    shift = None if shift is None else np.ascontiguousarray(shift, dtype=np.float32)
    cptr, _keep = _cfg(cfg)
    xl = to_l16(x)
    rl = None if residual is None else to_l16(residual)
    rc = lib().poco_op_conv2d(fptr(xl), B, H, W, Cin, fptr(weight), fptr(scale), fptr(shift), Cout, ks,
                              stride, fptr(rl), int(relu), fptr(out), cptr, current_stream())
    check(rc, "poco_op_conv2d")
    return from_l16(out)


def bench_conv2d(x: torch.Tensor, weight: np.ndarray, stride=1, cfg=None, iters=20):
    B, H, W, Cin = x.shape
    Cout, _, ks, _ = weight.shape
    pad = (ks - 1) // 2
    Ho = (H + 2 * pad - ks) // stride + 1
    Wo = (W + 2 * pad - ks) // stride + 1
    out = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    ms = C.c_float(0)
    used = (C.c_int * 7)()
    cptr, _keep = _cfg(cfg)
    rc = lib().poco_bench_conv2d(fptr(x), B, H, W, Cin, fptr(weight), Cout, ks, stride, fptr(out), cptr,
                                 iters, C.byref(ms), used, current_stream())
    check(rc, "poco_bench_conv2d")
    flops = 2.0 * B * Ho * Wo * Cout * Cin * ks * ks
    return ms.value, flops / (ms.value * 1e-3) / 1e12, tuple(used)


def part_attention(feat_nchw: torch.Tensor, heat_nchw: torch.Tensor) -> torch.Tensor:
    """KeypointAttention as the PARE head uses it: feat [B,C,H,W], heat [B,24,H,W] (the 24 part maps, no background channel)
    -> [B,C,24].  Channels are padded to the library's L16 layout here (tests only; the engine never leaves L16)."""
    B, Cc, H, W = feat_nchw.shape
    assert heat_nchw.shape == (B, 24, H, W) and feat_nchw.is_cuda
    C16 = (Cc + 15) // 16 * 16
    f = torch.zeros((B, H, W, C16), device=feat_nchw.device)
    f[..., :Cc] = feat_nchw.permute(0, 2, 3, 1)
    h = torch.zeros((B, H, W, 32), device=feat_nchw.device)
    h[..., 1:25] = heat_nchw.permute(0, 2, 3, 1)            # channel 0 = background, skipped by the kernel
    fl, hl = to_l16(f), to_l16(h)
    out = torch.empty((B, C16, 24), device=feat_nchw.device)
    check(lib().poco_op_part_attention(fptr(hl), 32, fptr(fl), C16, B, H, W, fptr(out), current_stream()),
          "poco_op_part_attention")
    return out[:, :Cc]


def lc2d_pose(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """x [B,128,24] , w [6,128,24] -> [B,24,6] (LocallyConnected2d 128 -> 6 per joint)."""
    B = x.shape[0]
    assert x.shape[1:] == (128, 24) and w.shape == (6, 128, 24) and x.is_cuda
    out = torch.empty((B, 24, 6), device=x.device)
    check(lib().poco_op_lc2d_pose(fptr(x.contiguous()), fptr(w.contiguous()), fptr(out), B, current_stream()), "poco_op_lc2d_pose")
    return out


def rot6d(x: torch.Tensor) -> torch.Tensor:
    """x [B,24,6] (each row = a 3x2 matrix, row-major as in the reference) -> rotation matrices [B,24,3,3]."""
    B = x.shape[0]
    assert x.shape[1:] == (24, 6) and x.is_cuda
    out = torch.empty((B, 24, 3, 3), device=x.device)
    check(lib().poco_op_rot6d(fptr(x.contiguous()), fptr(out), B, current_stream()), "poco_op_rot6d")
    return out
