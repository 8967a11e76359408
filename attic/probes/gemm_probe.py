"""Best configuration per ALG for the 1x1 conv shapes of ResNet-50 / HRNet layer1 (B = 64)."""
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
SHAPES = [(64, 56, 56, 64, 256, 1, 1), (64, 56, 56, 256, 64, 1, 1), (64, 56, 56, 256, 128, 1, 1), (64, 28, 28, 128, 512, 1, 1),
          (64, 28, 28, 512, 128, 1, 1), (64, 28, 28, 512, 256, 1, 1), (64, 14, 14, 256, 1024, 1, 1), (64, 14, 14, 1024, 256, 1, 1),
          (64, 14, 14, 1024, 512, 1, 1), (64, 7, 7, 512, 2048, 1, 1), (64, 7, 7, 2048, 512, 1, 1), (64, 56, 56, 256, 512, 1, 2),
          (64, 28, 28, 512, 1024, 1, 2), (64, 14, 14, 1024, 2048, 1, 2), (64, 7, 7, 1024, 2048, 1, 1)]
for (B, H, W, Cin, Cout, ks, st) in SHAPES:
    cands = tune.candidates(B, H, W, Cin, Cout, ks, st)
    flat = (C.c_int * (7 * len(cands)))(*[v for c in cands for v in c])
    ms = (C.c_float * len(cands))()
    check(L.poco_tune_conv(B, H, W, Cin, Cout, ks, st, flat, len(cands), 10, ms, None), "tune")
    byalg = {}
    for t, c in sorted((ms[i], cands[i]) for i in range(len(cands)) if ms[i] > 0):
        byalg.setdefault(c[6], (t, c))
    Ho = (H - 1) // st + 1
    fl = 2.0 * B * Ho * Ho * Cin * Cout
    print(f"{H}x{W} {Cin}->{Cout} s{st}: " + "  ".join(f"ALG{a}: {t*1e3:.1f}us ({fl/t/1e9:.0f} TF) {c}" for a, (t, c) in sorted(byalg.items())))
