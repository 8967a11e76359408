"""Probe for a GEMM-formulated Winograd on the small planes (round 3): how fast do the existing 1x1 GEMM kernels run the
position GEMMs  M_xi[Cout][tiles] = U_xi[Cout][Cin] . V_xi[Cin][tiles]  if V / M are staged in memory?  A 1x1 conv over
(positions x images) "images" of (tiles per image) pixels has the same operand traffic and MFMA work.

    python tools/wino_gemm_probe.py
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import tune  # noqa: E402
from poco_amd._lib import lib  # noqa: E402

PEAK = 157.3
L = lib()
# name, positions, tiles/image (as h x w), Cin, Cout, crops
CASES = [("14x14 192->192 F(4x4): 36 pos x 16 tiles", 36, (4, 4), 192, 192, 64),
         ("14x14 192->192 F(2x2): 16 pos x 49 tiles", 16, (7, 7), 192, 192, 64),
         ("7x7 384->384 F(2x2): 16 pos x 16 tiles", 16, (4, 4), 384, 384, 64),
         ("7x7 384->384 F(4x4): 36 pos x 4 tiles", 36, (2, 2), 384, 384, 64),
         ("28x28 96->96 F(4x4): 36 pos x 49 tiles", 36, (7, 7), 96, 96, 64)]
for name, npos, (th, tw), Cin, Cout, crops in CASES:
    B = npos * crops
    res = [r for r in tune.solo_times(L, B, th, tw, Cin, Cout, 1, 1, iters=10) if r[0] > 0]
    res.sort()
    gf = 2.0 * B * th * tw * Cin * Cout / 1e9
    print(f"{name}: {gf:.2f} GFLOP executed")
    for ms, cfg in res[:5]:
        print(f"   {ms*1e3:8.1f} us  {gf/ms/1e3:6.1f} TF ({gf/ms/1e3/PEAK:.2f})  cfg {cfg}")
