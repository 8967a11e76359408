"""ALG 11 (conv_wino4g.hip: F(4x4,3x3) as 36 position GEMMs) against the other kernels on the 7x7 shapes (solo timings)."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import tune  # noqa: E402
from poco_amd._lib import lib  # noqa: E402

L = lib()
table = tune.load_table()
for B, H, W, Cin, Cout in [(64, 7, 7, 384, 384), (128, 7, 7, 384, 384), (32, 7, 7, 256, 256), (16, 7, 7, 384, 384), (1, 7, 7, 384, 384)]:
    res = sorted(r for r in tune.solo_times(L, B, H, W, Cin, Cout, 3, 1, iters=10) if r[0] > 0)
    gf = 2.0 * 9 * B * H * W * Cin * Cout / 1e9
    best = {}
    for ms, cfg in res:
        best.setdefault(cfg[6], []).append((ms, cfg))
    print(f"B={B} {H}x{W} {Cin}->{Cout}: {gf:.2f} GF; table {table.get(tune.shape_key(B, H, W, Cin, Cout, 3, 1))}")
    for alg, lst in sorted(best.items(), key=lambda kv: kv[1][0][0]):
        for ms, cfg in lst[:3 if alg == 11 else 1]:
            print(f"     ALG {alg:2d} {ms*1e3:7.1f} us {gf/ms:6.1f} TF {cfg}")
