"""Same-box A/B of ALG 13 builds on the dominant shapes, without and WITH a residual (conv1 / conv2 of a BasicBlock):
python tools/w4w_ab.py [B]  (run once per POCO_HIP_LIB)"""
import ctypes as C
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from poco_amd._lib import check, lib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
torch.cuda.set_device(0)
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
SHAPES = [((56, 56, 48, 48), [(1, 3, 2, 1, 8, 1, 13), (1, 3, 2, 1, 4, 0, 13)]),
          ((28, 28, 96, 96), [(1, 3, 2, 1, 16, 1, 13), (1, 3, 2, 1, 4, 0, 13)]),
          ((14, 14, 192, 192), [(2, 3, 2, 1, 16, 2, 13)]),
          ((56, 56, 64, 64), [(1, 2, 2, 1, 8, 1, 13)]),
          ((56, 56, 32, 32), [(1, 2, 2, 1, 8, 1, 13)]),
          ((56, 56, 480, 128), [(1, 3, 2, 1, 4, 0, 13)])]
for res in (0, 1):
    out = []
    for (H, W, Cin, Cout), cfgs in SHAPES:
        for cfg in cfgs:
            if res and Cin != Cout:
                out.append("-")
                continue
            # slot 0 is a throw-away copy (the first configuration of a call times ~10 % slow)
            arr = (C.c_int * 21)(*(cfg * 3))
            ms = (C.c_float * 3)()
            check(L.poco_tune_conv(B, H, W, Cin, Cout, 3, 1, arr, 3, -40 if res else 40, ms, None), "tune")
            out.append(f"{min(ms[1], ms[2]) * 1e3:.1f}")
    print(("res   " if res else "nores ") + " | ".join(out), flush=True)
