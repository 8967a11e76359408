"""Micro-benchmark of the MFMA conv kernel on the dominant HRNet shapes (B=64 by default).

    python tools/bench_conv.py [--batch 64] [--sweep]
Prints ms, TFLOP/s and fraction of the 157.3 TF fp32-MFMA peak per shape.
"""
import argparse
import itertools
import json
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import ops  # noqa: E402

PEAK = 157.3

SHAPES = [
    # H, W, Cin, Cout, ks, stride
    (56, 56, 32, 32, 3, 1), (28, 28, 64, 64, 3, 1), (14, 14, 128, 128, 3, 1), (7, 7, 256, 256, 3, 1),
    (56, 56, 48, 48, 3, 1), (28, 28, 96, 96, 3, 1), (14, 14, 192, 192, 3, 1), (7, 7, 384, 384, 3, 1),
    (56, 56, 64, 64, 3, 1), (56, 56, 64, 256, 1, 1), (56, 56, 256, 64, 1, 1),
    (56, 56, 256, 256, 3, 1), (56, 56, 480, 128, 3, 1), (112, 112, 64, 64, 3, 2),
    (7, 7, 1024, 2048, 1, 1), (14, 14, 512, 512, 3, 1),
]


def candidates(H, W, Cin, Cout, ks, stride):
    pad = (ks - 1) // 2
    Ho = (H + 2 * pad - ks) // stride + 1
    Wo = (W + 2 * pad - ks) // stride + 1
    nT = Cout // 16
    out = set()
    for MT, NT, WM, WN in itertools.product((4, 7, 13), (1, 2, 3, 4), (1, 2, 4, 8), (1, 2, 4, 8)):
        if WM * WN > 8 or WM * WN < 2 or (MT == 13 and NT > (2 if ks == 3 else 3)):
            continue
        if nT % NT and NT > nT:
            continue
        cap = WM * MT * 16
        for R in range(1, Ho + 1):
            if R * Wo > cap:
                break
            NI = cap // (R * Wo) if R == Ho or True else 1
            for ni in {1, NI}:
                if ni < 1 or ni * R * Wo > cap:
                    continue
                eff = ni * R * Wo / cap
                if eff < 0.8:
                    continue
                pr = (R - 1) * stride + ks
                pw = (Wo - 1) * stride + ks
                lds = 4 * ((ni * pr * pw + 15) // 16 * 16) * 16
                if lds > 150 * 1024:
                    continue
                out.add((MT, NT, WM, WN, R, ni))
    return sorted(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--sweep", action="store_true")
    ap.add_argument("--out", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)
    results = []
    for (H, W, Cin, Cout, ks, stride) in SHAPES:
        x = torch.from_numpy(rng.standard_normal((args.batch, H, W, Cin)).astype(np.float32)).to(dev)
        w = (rng.standard_normal((Cout, Cin, ks, ks)) / np.sqrt(Cin * ks * ks)).astype(np.float32)
        ms, tf, used = ops.bench_conv2d(x, w, stride)
        line = dict(shape=[H, W, Cin, Cout, ks, stride], ms=round(ms, 4), tflops=round(tf, 1),
                    frac=round(tf / PEAK, 3), cfg=list(used))
        if args.sweep:
            best = (ms, used)
            for cfg in candidates(H, W, Cin, Cout, ks, stride):
                try:
                    m2, t2, _ = ops.bench_conv2d(x, w, stride, cfg=cfg, iters=10)
                except Exception as e:  # invalid cfg
                    continue
                if m2 < best[0]:
                    best = (m2, cfg)
            flops = tf * ms
            line["best_ms"] = round(best[0], 4)
            line["best_cfg"] = list(best[1])
            line["best_frac"] = round(flops / best[0] / PEAK, 3)
        print(json.dumps(line), flush=True)
        results.append(line)
    if args.out:
        Path(args.out).write_text(json.dumps(results, indent=1))


if __name__ == "__main__":
    main()
