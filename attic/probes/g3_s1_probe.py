"""ALG 10 (gather GEMM) on the stride-1 3x3 shapes of HRNet-W32 at B = 32 against the tuned Winograd entries (solo)."""
import os
import sys
from pathlib import Path

os.environ["POCO_TUNE_G3_S1"] = "1"
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import tune  # noqa: E402
from poco_amd._lib import lib  # noqa: E402

L = lib()
table = tune.load_table()
for B, H, W, Cin, Cout in [(32, 56, 56, 32, 32), (32, 28, 28, 64, 64), (32, 14, 14, 128, 128), (32, 7, 7, 256, 256),
                           (64, 56, 56, 48, 48), (64, 7, 7, 384, 384), (32, 56, 56, 64, 64), (1, 56, 56, 48, 48), (16, 56, 56, 48, 48),
                           (16, 14, 14, 192, 192), (16, 7, 7, 384, 384)]:
    res = sorted(r for r in tune.solo_times(L, B, H, W, Cin, Cout, 3, 1, iters=10) if r[0] > 0)
    gf = 2.0 * 9 * B * H * W * Cin * Cout / 1e9
    best = {}
    for ms, cfg in res:
        best.setdefault(cfg[6], (ms, cfg))
    print(f"B={B} {H}x{W} {Cin}->{Cout}: {gf:.2f} GF; table {table.get(tune.shape_key(B, H, W, Cin, Cout, 3, 1))}")
    for alg, (ms, cfg) in sorted(best.items(), key=lambda kv: kv[1][0]):
        print(f"     ALG {alg:2d} {ms*1e3:7.1f} us {gf/ms:6.1f} TF {cfg}")
