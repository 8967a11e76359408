"""Fixed vs per-slice cost of stride-2 3x3 conv configurations (like wino_slope.py)."""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from poco_amd._lib import check, lib  # noqa: E402
from poco_amd import tune  # noqa: E402
torch.cuda.set_device(0)
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
for (B, H, W, Cout, ks, st) in [(64, 56, 56, 144, 3, 2), (64, 28, 28, 192, 3, 2), (64, 56, 56, 256, 1, 1)]:
    best = {}
    for Cin in (48, 96, 192):
        cands = tune.candidates(B, H, W, Cin, Cout, ks, st)
        flat = (C.c_int * (7 * len(cands)))(*[v for c in cands for v in c])
        ms = (C.c_float * len(cands))()
        check(L.poco_tune_conv(B, H, W, Cin, Cout, ks, st, flat, len(cands), 10, ms, None), "tune")
        r = sorted((ms[i], cands[i]) for i in range(len(cands)) if ms[i] > 0)
        best[Cin] = r[:3]
        byalg = {}
        for t, c in r:
            byalg.setdefault(c[6], (t, c))
        print(f"{H}x{W} {Cin}->{Cout} k{ks}s{st}: " + "  ".join(f"ALG{a}: {t*1e3:.1f}us {c}" for a, (t, c) in sorted(byalg.items())))
    flops = lambda Cin: 2.0 * B * ((H + 2 * ((ks - 1) // 2) - ks) // st + 1) ** 2 * Cout * Cin * ks * ks
    t48, t192 = best[48][0][0], best[192][0][0]
    per16 = (t192 - t48) / 9
    print(f"   per 16-ch slice {per16*1e3:.2f} us (MFMA-ideal {flops(16)/157.3e12*1e6:.2f} us), fixed {(t48 - 3*per16)*1e3:.1f} us")
