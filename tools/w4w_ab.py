"""Same-box A/B of ALG 13 builds on the dominant shapes: python tools/w4w_ab.py [B]  (run once per POCO_HIP_LIB)"""
import sys
from pathlib import Path
import numpy as np
import torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from poco_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
SHAPES = [((56, 56, 48, 48), [(1, 3, 2, 1, 8, 1, 13), (1, 3, 2, 1, 4, 0, 13)]),
          ((28, 28, 96, 96), [(1, 3, 2, 1, 16, 1, 13), (1, 3, 2, 1, 4, 0, 13)]),
          ((14, 14, 192, 192), [(2, 3, 2, 1, 16, 2, 13)]),
          ((56, 56, 64, 64), [(1, 2, 2, 1, 8, 1, 13)]),
          ((56, 56, 32, 32), [(1, 2, 2, 1, 8, 1, 13)]),
          ((56, 56, 480, 128), [(1, 3, 2, 1, 4, 0, 13)])]
res = []
for (H, W, Cin, Cout), cfgs in SHAPES:
    x = torch.randn(B, H, W, Cin, device=dev)
    w = (np.random.default_rng(0).standard_normal((Cout, Cin, 3, 3)) / np.sqrt(9 * Cin)).astype(np.float32)
    for cfg in cfgs:
        try:
            ts = [ops.bench_conv2d(x, w, 1, cfg=cfg, iters=40)[0] * 1e3 for _ in range(3)]
            res.append(f"{min(ts):.1f}")
        except Exception as e:
            res.append("refused")
print(" | ".join(res), flush=True)
