"""Where the waves of conv_wino2_kernel spend their shader clocks (s_memtime sums per phase, per wave).

Needs the instrumented build of the Winograd TU (the regular library has no probe code):
    tools/build_exp.sh conv_wino.hip WINO_TRACE 1
    POCO_HIP_LIB=$PWD/poco_amd/lib/exp/libpoco_hip_WINO_TRACE_1.so python tools/wino_trace.py
Also prints the shader clock the kernel actually ran at (s_memtime ticks / 100 MHz s_memrealtime ticks)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from poco_amd._lib import check, lib  # noqa: E402

torch.cuda.set_device(0)
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
L.poco_debug_wino_trace.argtypes = [C.c_void_p, C.c_size_t]
NAMES = ["tile-start wait+barrier", "slice body", "own-DMA wait", "slice barrier", "end-of-tile stage", "kernel"]
CASES = [((64, 56, 56, 48, 48), (1, 3, 4, 2, 4, 1, 4)), ((64, 56, 56, 192, 48), (1, 3, 4, 2, 4, 1, 4)),
         ((64, 28, 28, 96, 96), (1, 3, 4, 2, 14, 1, 4)), ((64, 14, 14, 192, 192), (1, 3, 4, 2, 14, 1, 4))]
for (B, H, W, Cin, Cout), cfg in CASES:
    flat = (C.c_int * 7)(*cfg)
    ms = (C.c_float * 1)()
    check(L.poco_tune_conv(B, H, W, Cin, Cout, 3, 1, flat, 1, 5, ms, None), "tune")
    buf = np.zeros((512, 8, 8), dtype=np.uint64)
    assert L.poco_debug_wino_trace(buf.ctypes.data, buf.size) == 0
    nb = int((buf[:, 0, 5] > 0).sum())
    t = buf[:nb].astype(np.float64)
    print(f"{H}x{W} {Cin}->{Cout} cfg={cfg}: {ms[0]*1e3:.1f} us, {nb} blocks traced; clocks per wave (mean over blocks):")
    for half, sl in (("lower", slice(0, 4)), ("upper", slice(4, 8))):
        m = t[:, sl, :].mean(axis=(0, 1))
        tot = m[5]
        print(f"  {half} half: " + "  ".join(f"{NAMES[k]} {m[k]:.0f} ({100*m[k]/tot:.0f}%)" for k in range(6)))
    print(f"  shader clock: {t[:, 0, 5].mean() / t[:, 0, 7].mean() * 100:.0f} MHz (s_memtime / s_memrealtime)")
    st = t[:, 0, 6]
    print(f"  block start skew: {st.max()-st.min():.0f} clk; kernel clocks min/max over blocks {t[:,0,5].min():.0f}/{t[:,0,5].max():.0f}")
