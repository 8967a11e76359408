"""Time a few 1x1 GEMM shapes with fixed ALG 6 configurations (used with the G1_EXP probe builds, tools/build_exp.sh)."""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from poco_amd._lib import check, lib  # noqa: E402
torch.cuda.set_device(0)
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
import os
NIS = [int(x) for x in os.environ.get("G1_NI", "1").split(",")]
BASE = [((64, 14, 14, 1024, 256, 1, 1), [(7, 2, 1, 1, 3), (7, 2, 2, 2, 2), (7, 4, 1, 1, 2), (4, 4, 2, 2, 2)]),
        ((64, 14, 14, 256, 1024, 1, 1), [(7, 4, 1, 1, 2), (7, 4, 2, 2, 2), (7, 2, 2, 2, 3)]),
        ((64, 28, 28, 512, 256, 1, 1), [(7, 4, 2, 2, 2), (7, 4, 1, 1, 2), (7, 2, 2, 2, 2)]),
        ((64, 56, 56, 256, 256, 1, 1), [(7, 4, 2, 2, 2), (7, 4, 1, 1, 2), (7, 4, 1, 1, 3), (7, 2, 2, 2, 2)])]
CASES = [(sh, [c + (ni, 6) for c in cf for ni in NIS] +
          [(mt, nt, wm, wn, 1, 1, 9) for (mt, nt) in ((7, 4), (7, 2), (8, 2), (4, 4)) for (wm, wn) in ((1, 1), (2, 2), (1, 4), (2, 1), (4, 1), (4, 2), (2, 4))])
         for sh, cf in BASE]
for shape, cfgs in CASES:
    B, H, W, Cin, Cout, ks, st = shape
    flat = (C.c_int * (7 * len(cfgs)))(*[v for c in cfgs for v in c])
    ms = (C.c_float * len(cfgs))()
    check(L.poco_tune_conv(*shape, flat, len(cfgs), 20, ms, None), "tune")
    fl = 2.0 * B * H * W * Cin * Cout
    print(f"{H}x{W} {Cin}->{Cout}: " + "  ".join(f"{c[:6]}: {ms[i]*1e3:.1f}us ({fl/ms[i]/1e9:.0f} TF)" for i, c in enumerate(cfgs)))
