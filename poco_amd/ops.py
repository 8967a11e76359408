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
