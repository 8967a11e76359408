"""Split a conv launch time into fixed (prologue/epilogue/launch) and per-K-loop cost by timing
POCO_CONV_REPEAT=1 vs 3 in two processes.  usage: python tools/conv_fixed_cost.py  (runs itself)"""
import ctypes as C
import json
import os
import subprocess
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
SHAPES = [
    ((64, 56, 56, 48, 48, 3, 1), [(7, 3, 8, 1, 16, 1, 0), (7, 3, 4, 1, 8, 1, 1), (7, 3, 4, 1, 8, 1, 2), (7, 3, 2, 1, 4, 1, 2), (7, 3, 4, 1, 4, 2, 2)]),
    ((64, 28, 28, 96, 96, 3, 1), [(7, 3, 4, 1, 7, 2, 1), (7, 3, 4, 1, 7, 2, 2), (7, 3, 2, 2, 7, 1, 2)]),
    ((64, 14, 14, 192, 192, 3, 1), [(4, 3, 4, 1, 7, 2, 1), (4, 3, 4, 1, 7, 2, 2)]),
    ((64, 7, 7, 384, 384, 3, 1), [(7, 1, 4, 1, 7, 9, 1), (7, 1, 4, 1, 7, 9, 2)]),
]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from poco_amd._lib import lib, check
    import torch
    torch.cuda.set_device(0)
    L = lib()
    L.poco_tune_conv.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int), C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p]
    out = {}
    for shp, cfgs in SHAPES:
        flat = (C.c_int * (7 * len(cfgs)))(*[v for c in cfgs for v in c])
        ms = (C.c_float * len(cfgs))()
        check(L.poco_tune_conv(*shp, flat, len(cfgs), 30, ms, None), "tune")
        out[str(shp)] = [ms[i] for i in range(len(cfgs))]
    print("RESULT" + json.dumps(out))
else:
    res = {}
    for rep in (0, 1, 3):
        env = dict(os.environ, POCO_CONV_REPEAT=str(rep))
        o = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout
        res[rep] = json.loads([l for l in o.splitlines() if l.startswith("RESULT")][0][6:])
    for shp, cfgs in SHAPES:
        B, H, W, Cin, Cout, ks, st = shp
        fl = 2.0 * B * (H // st) * (W // st) * Cin * Cout * ks * ks
        for i, cfg in enumerate(cfgs):
            t1, t3 = res[1][str(shp)][i], res[3][str(shp)][i]
            loop = (t3 - t1) / 2
            fixed = t1 - loop
            t0 = res[0][str(shp)][i]
            print(f"{shp} cfg={cfg}: rep0={t0*1e3:.1f}us t1={t1*1e3:.1f}us loop={loop*1e3:.1f}us ({fl/loop/1e9:.1f} TF asymptotic) fixed={fixed*1e3:.1f}us ({fl/t1/1e9:.1f} TF actual)")
