"""Per-16-channel-slice cost vs fixed cost of a conv config: time Cin = 48, 96, 192 at fixed Cout."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from poco_amd._lib import check, lib  # noqa: E402

torch.cuda.set_device(0)
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
CASES = [((64, 56, 56, 48, 3, 1), [(1, 1, 4, 3, 4, 1, 3), (1, 1, 4, 2, 4, 1, 4), (1, 2, 4, 2, 4, 1, 4), (1, 3, 4, 2, 4, 1, 4), (1, 3, 2, 2, 2, 1, 4), (7, 3, 4, 1, 8, 1, 1)]),
         ((64, 14, 14, 192, 3, 1), [(1, 1, 4, 1, 14, 1, 3), (1, 1, 4, 2, 14, 1, 4), (1, 2, 4, 2, 14, 1, 4), (1, 3, 4, 2, 14, 1, 4), (4, 3, 4, 1, 7, 2, 2)])]
for (B, H, W, Cout, ks, st), cfgs in CASES:
    for cfg in cfgs:
        ts = []
        for Cin in (48, 96, 192, 384):
            flat = (C.c_int * 7)(*cfg)
            ms = (C.c_float * 1)()
            check(L.poco_tune_conv(B, H, W, Cin, Cout, ks, st, flat, 1, 20, ms, None), "tune")
            ts.append(ms[0] * 1e3)
        per16 = (ts[3] - ts[1]) / 18
        print(f"{H}x{W} Cout={Cout} cfg={cfg}: t(Cin=48,96,192,384)={['%.1f' % t for t in ts]} us  per-slice {per16:.2f} us  fixed {ts[1]-6*per16:.1f} us")
