"""Re-time every tuned-table entry that runs on ALG 7 (F(4x4) Winograd) against the ALG 8 candidates of the same shape
(specialised waves) and keep the faster one.  python tools/retune_w4.py [--write]"""
import ctypes as C
import json
import re
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch  # noqa: E402
from poco_amd import tune  # noqa: E402
from poco_amd._lib import check, lib  # noqa: E402
torch.cuda.set_device(0)
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
full = json.loads(tune.TABLE.read_text())
changed = 0
for key, ent in sorted(full.items()):
    cfg = ent["cfg"]
    if cfg[6] not in (7, 8):
        continue
    B, H, W, Cin, Cout, ks, st = map(int, re.fullmatch(r"(\d+)x(\d+)x(\d+)x(\d+)x(\d+)k(\d+)s(\d+)", key).groups())
    cands = [tuple(cfg)] + [c for c in tune.candidates(B, H, W, Cin, Cout, ks, st) if c[6] in (7, 8) and tuple(c) != tuple(cfg)]
    flat = (C.c_int * (7 * len(cands)))(*[v for c in cands for v in c])
    ms = (C.c_float * len(cands))()
    check(L.poco_tune_conv(B, H, W, Cin, Cout, ks, st, flat, len(cands), 20, ms, None), "tune")
    res = sorted((ms[i], cands[i]) for i in range(len(cands)) if ms[i] > 0)
    best_ms, best = res[0]
    cur_ms = ms[0]
    b7 = min((m for m, c in res if c[6] == 7), default=0.0)
    b8 = min((m for m, c in res if c[6] == 8), default=0.0)
    print(f"{key:28s} x{ent.get('uses', 0):3d}  current {tuple(cfg)} {cur_ms*1e3:7.1f} us | best ALG7 {b7*1e3:7.1f} | best ALG8 {b8*1e3:7.1f} -> {best}", flush=True)
    if tuple(best) != tuple(cfg) and best_ms < 0.985 * cur_ms:
        ent["cfg"] = list(best); ent["ms"] = round(float(best_ms), 5)
        fl = 2.0 * B * H * W * Cout * Cin * 9
        ent["tflops"] = round(fl / best_ms / 1e9, 1)
        changed += 1
print("entries changed:", changed)
if "--write" in sys.argv:
    tune.TABLE.write_text(json.dumps(full, indent=0, sort_keys=True))
    print("wrote", tune.TABLE)
