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
    out = torch.empty((B, Ho, Wo, Cout), device=x.device, dtype=torch.float32)
    weight = np.ascontiguousarray(weight, dtype=np.float32)
    scale = None if scale is None else np.ascontiguousarray(scale, dtype=np.float32)
    shift = None if shift is None else np.ascontiguousarray(shift, dtype=np.float32)
    cptr, _keep = _cfg(cfg)
    rc = lib().poco_op_conv2d(fptr(x), B, H, W, Cin, fptr(weight), fptr(scale), fptr(shift), Cout, ks,
                              stride, fptr(residual), int(relu), fptr(out), cptr, current_stream())
    check(rc, "poco_op_conv2d")
    return out


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
