"""ALG 10 (gemm3x3.hip) against the best LDS-staged configuration per stride-2 3x3 shape (solo timings, 64 crops).

    python tools/g3_probe.py [--batch 64] [--quick]
"""
import argparse
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import tune  # noqa: E402
from poco_amd._lib import lib  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--quick", action="store_true")
args = ap.parse_args()
L = lib()
SHAPES = [(56, 56, 48, 144), (28, 28, 96, 192), (14, 14, 512, 1024), (56, 56, 128, 256), (28, 28, 256, 512), (28, 28, 96, 288),
          (56, 56, 48, 192), (14, 14, 192, 384), (28, 28, 48, 192), (56, 56, 256, 96), (112, 112, 64, 64), (14, 14, 96, 384),
          (14, 14, 48, 384), (56, 56, 128, 128), (28, 28, 256, 256), (14, 14, 512, 512), (56, 56, 32, 64), (28, 28, 64, 128)]
if args.quick:
    SHAPES = SHAPES[:4]
table = tune.load_table()
for H, W, Cin, Cout in SHAPES:
    B = args.batch
    res = [r for r in tune.solo_times(L, B, H, W, Cin, Cout, 3, 2, iters=8) if r[0] > 0]
    gf = 2.0 * 9 * B * ((H + 1) // 2) * ((W + 1) // 2) * Cin * Cout / 1e9
    old = sorted(r for r in res if r[1][6] != 10)[:1]
    new = sorted(r for r in res if r[1][6] == 10)[:3]
    cur = table.get(tune.shape_key(B, H, W, Cin, Cout, 3, 2))
    print(f"{H}x{W} {Cin}->{Cout} s2: {gf:.2f} GF | best staged {old[0][0]*1e3:7.1f} us {gf/old[0][0]:6.1f} TF {old[0][1]} | table {cur}")
    for ms, cfg in new:
        print(f"      ALG 10 {ms*1e3:7.1f} us {gf/ms:6.1f} TF {cfg}  ({old[0][0]/ms:.2f}x)")
