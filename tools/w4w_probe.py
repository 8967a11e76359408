"""Timing probes of ALG 13 (W4W_EXP builds, results are garbage): one long-K shape, us per launch and per slice"""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import ops  # noqa: E402
dev = torch.device("cuda:0")
for (B, H, W, Cin, Cout), cfg in [((64, 14, 14, 192, 192), (2, 3, 2, 1, 16, 2, 13)), ((64, 14, 14, 768, 192), (2, 3, 2, 1, 16, 2, 13))]:
    x = torch.randn(B, H, W, Cin, device=dev)
    w = (np.random.default_rng(0).standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    ts = [ops.bench_conv2d(x, w, 1, cfg=cfg, iters=30)[0] * 1e3 for _ in range(3)]
    print(f"{H}x{W} {Cin}->{Cout} {cfg}: {min(ts):.1f} us = {min(ts) / (Cin / 4):.3f} us / slice", flush=True)
