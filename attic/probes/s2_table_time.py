"""Time the tuned (shape, cfg) entries with stride 2 (or any key substring) of one batch size: python tools/s2_table_time.py 64 [substr]
(run it under POCO_HIP_LIB=... to compare two builds of the library on one box)."""
import ctypes as C
import json
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from poco_amd._lib import check, lib  # noqa: E402
torch.cuda.set_device(0)
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
B = int(sys.argv[1]); sub = sys.argv[2] if len(sys.argv) > 2 else "s2"
t = json.loads((Path(__file__).resolve().parent.parent / "poco_amd" / "tuned" / "gfx950.json").read_text())
tot = 0.0
for k in sorted(t):
    if not k.startswith(f"{B}x") or sub not in k:
        continue
    dims, rest = k.split("k")
    _, H, W, Cin, Cout = map(int, dims.split("x"))
    ks, st = int(rest[0]), int(rest[2])
    cfg = t[k]["cfg"]
    flat = (C.c_int * 7)(*cfg); ms = (C.c_float * 1)()
    check(L.poco_tune_conv(B, H, W, Cin, Cout, ks, st, flat, 1, 20, ms, None), "tune")
    uses = t[k].get("uses", 1)
    tot += ms[0] * uses
    print(f"{k:28s} {cfg}  {ms[0]*1e3:7.1f} us  x{uses}  (table {t[k].get('ms', 0)*1e3:7.1f})")
print(f"sum over uses: {tot:.3f} ms")
