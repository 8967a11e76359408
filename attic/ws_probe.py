"""ALG 15 (weight-stationary direct 3x3 conv, csrc/conv3x3ws.hip) against the tuned table's entry: solo time per shape (warm slots).
    python tools/ws_probe.py [B]"""
import ctypes as C
import json
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd._lib import check, lib  # noqa: E402
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
table = json.loads((Path(__file__).resolve().parent.parent / "poco_amd" / "tuned" / "gfx950.json").read_text())
L = lib()
L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
for (H, W, Cc, cands) in [(56, 56, 32, [(7, 1, 2, 2, 4, 1, 15), (7, 1, 4, 2, 8, 1, 15), (7, 1, 1, 2, 2, 1, 15), (7, 1, 3, 2, 6, 1, 15)]),
                          (28, 28, 64, [(7, 1, 1, 4, 4, 1, 15), (7, 1, 1, 4, 3, 1, 15), (7, 1, 1, 4, 2, 1, 15)])]:
    key = f"{B}x{H}x{W}x{Cc}x{Cc}k3s1"
    tcfg = tuple(table[key]["cfg"]) if key in table else (0,) * 7
    run = [tcfg, tcfg] + cands + [tcfg]
    flat = (C.c_int * (7 * len(run)))(*[v for c in run for v in c])
    ms = (C.c_float * len(run))()
    check(L.poco_tune_conv(B, H, W, Cc, Cc, 3, 1, flat, len(run), 30, ms, None), "poco_tune_conv")
    fl = 2.0 * B * H * W * Cc * Cc * 9
    print(f"{H}x{W} {Cc}->{Cc} B={B}: table {tcfg} {min(ms[1], ms[len(run) - 1]) * 1e3:.1f} us | " +
          "  ".join(f"{c[:5]}: {ms[i + 2] * 1e3:.1f} us ({fl / ms[i + 2] / 1e9:.0f} TF)" for i, c in enumerate(cands)), flush=True)
