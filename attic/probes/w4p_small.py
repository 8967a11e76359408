"""ALG 8 / 7 / 4 on the small planes (14x14 192->192, 7x7 384->384) at 64 crops: is F(4x4) with specialised waves worth offering there?"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from poco_amd._lib import check, lib  # noqa: E402
torch.cuda.set_device(0)
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
CASES = [((64, 14, 14, 192, 192), [(1, 3, 2, 4, 16, 2, 8), (1, 3, 2, 4, 16, 1, 8), (1, 2, 2, 4, 16, 2, 8), (1, 2, 2, 4, 16, 1, 8), (1, 1, 2, 4, 16, 2, 8), (1, 3, 2, 4, 16, 2, 7), (1, 3, 4, 2, 14, 1, 4)]),
         ((64, 7, 7, 384, 384), [(1, 3, 2, 4, 8, 8, 8), (1, 3, 2, 4, 8, 4, 8), (1, 2, 2, 4, 8, 8, 8), (1, 2, 2, 4, 8, 4, 8), (1, 1, 2, 4, 8, 8, 8), (1, 3, 4, 2, 8, 4, 4), (1, 2, 4, 2, 8, 4, 4)])]
for (B, H, W, Cin, Cout), cfgs in CASES:
    for cfg in cfgs:
        flat = (C.c_int * 7)(*cfg); ms = (C.c_float * 1)()
        rc = L.poco_tune_conv(B, H, W, Cin, Cout, 3, 1, flat, 1, 20, ms, None)
        fl = 2.0 * B * H * W * Cin * Cout * 9
        print(f"{H}x{W} {Cin}->{Cout} cfg={cfg}: " + (f"{ms[0]*1e3:.1f} us  {fl/ms[0]/1e9:.0f} TF algorithmic" if rc == 0 and ms[0] > 0 else "refused"), flush=True)
