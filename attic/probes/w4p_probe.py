"""Slope (per 16 input channels) / fixed cost of one F(4x4) configuration: python tools/w4p_probe.py [alg]"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from poco_amd._lib import check, lib  # noqa: E402
torch.cuda.set_device(0)
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
alg = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for (B, H, W, Cout), cfg in [((64, 56, 56, 48), (1, 3, 2, 4, 8, 1, alg)), ((64, 28, 28, 96), (1, 3, 2, 4, 16, 1, alg))]:
    ts = []
    for Cin in (48, 96, 192):
        flat = (C.c_int * 7)(*cfg); ms = (C.c_float * 1)()
        check(L.poco_tune_conv(B, H, W, Cin, Cout, 3, 1, flat, 1, 20, ms, None), "tune")
        ts.append(ms[0] * 1e3)
    per16 = (ts[2] - ts[0]) / 9
    print(f"{H}x{W} Cout={Cout} cfg={cfg}: t(Cin=48,96,192)={['%.1f' % t for t in ts]} us  per 16 ch {per16:.2f} us  fixed {ts[0]-3*per16:.1f} us")
