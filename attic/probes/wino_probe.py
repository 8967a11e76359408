"""Time chosen Winograd (ALG 3) tile configs on the dominant shapes; prints effective TFLOP/s."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from poco_amd._lib import check, lib  # noqa: E402
from poco_amd.tune import candidates  # noqa: E402

torch.cuda.set_device(0)
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
for shp in [(64, 56, 56, 48, 48, 3, 1), (64, 28, 28, 96, 96, 3, 1), (64, 14, 14, 192, 192, 3, 1), (64, 7, 7, 384, 384, 3, 1)]:
    cands = [c for c in candidates(*shp) if c[6] in (3, 4)]
    flat = (C.c_int * (7 * len(cands)))(*[v for c in cands for v in c])
    ms = (C.c_float * len(cands))()
    check(L.poco_tune_conv(*shp, flat, len(cands), 10, ms, None), "tune")
    B, H, W, Cin, Cout, ks, st = shp
    fl = 2.0 * B * H * W * Cin * Cout * 9
    res = sorted((ms[i], cands[i]) for i in range(len(cands)) if ms[i] > 0)
    print(shp, "wino candidates:", len(cands))
    for t, c in res[:6]:
        print(f"   {t*1e3:7.1f} us  {fl/t/1e9:6.1f} TF eff  cfg={c}")
