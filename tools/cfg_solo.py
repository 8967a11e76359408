"""Solo time (warm slots) of explicit configurations of one conv shape: python tools/cfg_solo.py B H W Cin Cout ks stride "cfg" ["cfg" ...]"""
import ctypes as C, sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd._lib import check, lib  # noqa: E402
B, H, W, Cin, Cout, ks, st = map(int, sys.argv[1:8])
cfgs = [tuple(int(x) for x in c.split(",")) for c in sys.argv[8:]]
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
run = [cfgs[0]] + cfgs
flat = (C.c_int * (7 * len(run)))(*[v for c in run for v in c])
ms = (C.c_float * len(run))()
check(L.poco_tune_conv(B, H, W, Cin, Cout, ks, st, flat, len(run), 30, ms, None), "poco_tune_conv")
for i, c in enumerate(cfgs):
    print(f"{B}x{H}x{W} {Cin}->{Cout} k{ks}s{st} {c}: {ms[i + 1] * 1e3:.1f} us")
