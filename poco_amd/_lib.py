"""ctypes binding of libpoco_hip.so (the C-ABI in include/poco_hip.h).

The product path has NO fallback: if the HIP library is missing or a call fails, this module
raises.  (The CPU oracle under oracle/ is test infrastructure and is never imported from here.)
"""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parent
import os
# POCO_HIP_LIB: developer override (timing experiments with alternative builds); default = in-tree build
LIB_PATH = Path(os.environ.get("POCO_HIP_LIB") or (ROOT / "lib" / "libpoco_hip.so"))
HEADER = ROOT.parent / "include" / "poco_hip.h"

_lib = None


class PocoHipError(RuntimeError):
    pass


def header_symbols() -> list[str]:
    """Every function name declared in include/poco_hip.h."""
    txt = HEADER.read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(poco_[a-z0-9_]+)\s*\(", txt)))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise PocoHipError(
                f"{LIB_PATH} not found - build it with `python -m poco_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # torch ships its own libamdhip64; it must be the one HIP runtime in the process, so load it
        # before our library resolves its HIP symbols (otherwise two runtimes -> "no HIP device").
        import torch  # noqa: F401
        _lib = C.CDLL(str(LIB_PATH))
        _lib.poco_last_error.restype = C.c_char_p
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().poco_last_error().decode(errors="replace")
        raise PocoHipError(f"{what} failed (code {rc}): {msg}")


def fptr(t):
    """Device/host float pointer of a contiguous float32 torch tensor or numpy array (or None)."""
    if t is None:
        return C.c_void_p(0)
    import numpy as np
    if isinstance(t, np.ndarray):
        assert t.dtype == np.float32 and t.flags["C_CONTIGUOUS"]
        return C.c_void_p(t.ctypes.data)
    import torch
    assert t.dtype == torch.float32 and t.is_contiguous(), (t.dtype, t.is_contiguous())
    return C.c_void_p(t.data_ptr())


def iptr(t):
    if t is None:
        return C.c_void_p(0)
    import numpy as np
    if isinstance(t, np.ndarray):
        assert t.dtype == np.int32 and t.flags["C_CONTIGUOUS"]
        return C.c_void_p(t.ctypes.data)
    import torch
    assert t.dtype == torch.int32 and t.is_contiguous()
    return C.c_void_p(t.data_ptr())


def current_stream() -> C.c_void_p:
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
