"""Split-fp16 experiment (ALG 12) against the best fp32 configuration on ResNet-50's 1x1 shapes, with the HBM / MFMA floors."""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import tune  # noqa: E402
from poco_amd._lib import check, lib  # noqa: E402

L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
B = 64
for H, Cin, Cout in [(56, 64, 256), (56, 256, 64), (28, 128, 512), (28, 512, 128), (14, 256, 1024), (14, 1024, 256), (7, 512, 2048), (7, 2048, 512),
                     (56, 256, 128), (28, 512, 256), (14, 1024, 512)]:
    best = sorted(r for r in tune.solo_times(L, B, H, H, Cin, Cout, 1, 1, iters=10) if r[0] > 0)[0]
    cands = [(4, 4, 2, 2, 8, 1, 12), (4, 4, 2, 2, 2, 1, 12), (2, 4, 2, 2, 2, 1, 12), (4, 2, 2, 2, 2, 1, 12)]
    flat = (C.c_int * (7 * len(cands)))(*[v for c in cands for v in c])
    ms = (C.c_float * len(cands))()
    check(L.poco_tune_conv(B, H, H, Cin, Cout, 1, 1, flat, len(cands), 20, ms, None), "tune")
    gf = 2.0 * B * H * H * Cin * Cout / 1e9
    mb = 4.0 * B * H * H * (Cin + Cout) / 1e6
    print(f"{H}x{H} {Cin:4d}->{Cout:4d}: {gf:5.2f} GF {mb:6.1f} MB | floors: fp32-MFMA {gf/157.3*1e3:5.1f} us, HBM {mb/6.3:5.1f} us | fp32 best {best[0]*1e3:6.1f} us {best[1]} | "
          f"split tiled {ms[0]*1e3:6.1f}, direct " + " ".join(f"{ms[i]*1e3:.1f}" for i in (1, 2, 3)))
